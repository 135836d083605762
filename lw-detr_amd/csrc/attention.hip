// Fused softmax(Q K^T) V for gfx950 - window attention, global attention and the decoder's self-attention.
//
// Replaces the reference's materialised-score formulation (models/backbone/vit.py:130-137: a (B',12,N,N) tensor per
// block; models/attention.py:595-606) with a flash-style kernel: scores never leave registers.
//
// Layout contract (written by the QKV GEMM epilogue, gemm.hip): Q and K are (B, heads, Tp, hd) with Q pre-scaled by
// hd^-0.5 * log2(e); V is stored TRANSPOSED, (B, heads, hd, Tp), because P V contracts over keys and an MFMA operand
// needs its k-run contiguous per lane - the producing GEMM's accumulator layout makes that transposed store free.
// Tokens stay in the ViT's window-major order, so a window is a contiguous range of Twp rows and a global block is
// the whole image: the same kernel serves both (sequence geometry comes from the descriptor).
//
// Per wave: 16*QT queries. The score tile is computed TRANSPOSED, S^T = K Q^T (A operand = K rows, B operand = Q rows):
// the accumulator layout then gives every lane 4 keys x 1 query, which (a) makes the softmax row reduction two
// cross-lane steps (xor 16, 32) and (b) is already the B-operand layout of the second MFMA, O^T = V^T P^T - P goes from
// accumulator registers to MFMA operand with a type conversion only (no LDS, no shuffles). A 32-key step pairs the two
// 16-key tiles into one K=32 MFMA per 16 output channels; the k-slot -> key permutation this implies is applied to the
// V^T operand by loading two 8-byte runs per lane. Waves are independent (no LDS, no barriers): K / V^T tiles are
// re-read through L1/L2, which holds a whole (image, head) slice (1600 x 16 x 2 B x 2 = 100 KB at 640x640).
#include "common.h"

namespace {

template <typename T, int HD, int QT>
__global__ __launch_bounds__(256) void attn_kernel(const lwdetr_attn_desc p) {
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    constexpr int NC = HD >= 32 ? HD / 32 : 1;     // contraction chunks for Q K^T
    constexpr int DT = HD / 16;                    // 16-channel output tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, seq = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 16 * QT;
    if (q0 >= p.keys_per_seq) return;
    const int b = seq / p.seqs_per_img, w = seq - b * p.seqs_per_img;
    const long tok0 = (long)w * p.seq_tok_stride;
    const long bh = (long)b * p.heads + head;
    const T* __restrict__ Qb = (const T*)p.Q + (bh * p.Tp + tok0) * HD;
    const T* __restrict__ Kb = (const T*)p.K + (bh * p.Tp + tok0) * HD;
    const T* __restrict__ Vb = (const T*)p.VT + bh * HD * (long)p.Tp + tok0;
    const int nkeys = p.keys_per_seq;
    const int last = nkeys - 1;

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[q = l15][d-run of group g]
    V8 q8[QT][NC]; V4 q4[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int q = q0 + t * 16 + l15; q = q < last ? q : last;
        if (HD >= 32) {
#pragma unroll
            for (int c = 0; c < NC; ++c) q8[t][c] = *(const V8*)(Qb + (long)q * HD + c * 32 + g * 8);
        } else {
            q4[t] = *(const V4*)(Qb + (long)q * HD + g * 4);
        }
    }

    f32x4 o[QT][DT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -INFINITY; l_run[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const bool holes = p.sub_len < p.sub_stride;
    for (int k0 = 0; k0 < nkeys; k0 += 32) {
        // ---- S^T tiles: s[t][kt][r] = score(key k0 + kt*16 + g*4 + r, query q0 + t*16 + l15)
        f32x4 s[QT][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            int key = k0 + kt * 16 + l15; key = key < last ? key : last;
            if (HD >= 32) {
                V8 kf[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) kf[c] = *(const V8*)(Kb + (long)key * HD + c * 32 + g * 8);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < NC; ++c) a = Mma<T>::k32(kf[c], q8[t][c], a);
                    s[t][kt] = a;
                }
            } else {
                const V4 kf = *(const V4*)(Kb + (long)key * HD + g * 4);
#pragma unroll
                for (int t = 0; t < QT; ++t) s[t][kt] = Mma<T>::k16(kf, q4[t], f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
        // ---- mask keys beyond the sequence / pad rows inside it
        if (k0 + 32 > nkeys || holes) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + kt * 16 + g * 4 + r;
                    const bool ok = key < nkeys && (!holes || (key % p.sub_stride) < p.sub_len);
                    if (!ok) {
#pragma unroll
                        for (int t = 0; t < QT; ++t) s[t][kt][r] = -INFINITY;
                    }
                }
        }
        // ---- V^T fragments (A operand of O^T = V^T P^T): k-slots 0..3 <- keys k0+4g.., slots 4..7 <- keys k0+16+4g..
        V8 vf[DT];
        {
            int ka = k0 + g * 4, kb = k0 + 16 + g * 4;
            ka = ka + 3 < nkeys ? ka : (nkeys - 4);
            kb = kb + 3 < nkeys ? kb : (nkeys - 4);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const T* vrow = Vb + (long)(dt * 16 + l15) * p.Tp;
                const V4 lo = *(const V4*)(vrow + ka), hi = *(const V4*)(vrow + kb);
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[dt][e] = lo[e]; vf[dt][4 + e] = hi[e]; }
            }
        }
        // ---- online softmax (log2 domain) + P V
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = fmaxf(fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3])),
                             fmaxf(fmaxf(s[t][1][0], s[t][1][1]), fmaxf(s[t][1][2], s[t][1][3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[t], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
            m_run[t] = m_new;
            V8 pf;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(s[t][kt][r] - m_new);
                    psum += pv;
                    pf[kt * 4 + r] = from_f32<T>(pv);
                }
            l_run[t] = l_run[t] * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                o[t][dt] *= alpha;
                o[t][dt] = Mma<T>::k32(vf[dt], pf, o[t][dt]);
            }
        }
    }

    // ---- normalise and store: lane holds channels dt*16 + 4g .. +3 of query q0 + t*16 + l15
    T* __restrict__ out = (T*)p.out;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = l_run[t];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.f / l;
        const int q = q0 + t * 16 + l15;
        if (q < nkeys) {
            T* orow = out + ((long)b * p.Tp + tok0 + q) * p.ldo + head * HD;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                V4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = from_f32<T>(o[t][dt][r] * inv);
                *(V4*)(orow + dt * 16 + g * 4) = ov;
            }
        }
    }
}

template <typename T, int HD>
int launch(const lwdetr_attn_desc& p, hipStream_t st) {
    constexpr int QT = 2;
    const int units = (p.keys_per_seq + 16 * QT - 1) / (16 * QT);
    dim3 grid((units + 3) / 4, p.heads, p.B * p.seqs_per_img);
    const double nseq = (double)p.B * p.seqs_per_img;
    const double flops = 4.0 * nseq * p.heads * (double)p.keys_per_seq * p.keys_per_seq * HD;
    const double bytes = 4.0 * nseq * p.heads * p.keys_per_seq * HD * sizeof(T);
    const int kid = p.kind == 0 ? KID_ATTN_WINDOW : (p.kind == 1 ? KID_ATTN_GLOBAL : KID_ATTN_DECODER);
    ProfScope ps(kid, flops, bytes, st);
    hipLaunchKernelGGL((attn_kernel<T, HD, QT>), grid, dim3(256), 0, st, p);
    return lwdetr_check_launch();
}

template <typename T>
int dispatch_hd(const lwdetr_attn_desc& p, hipStream_t st) {
    switch (p.hd) {
        case 16: return launch<T, 16>(p, st);
        case 32: return launch<T, 32>(p, st);
        case 64: return launch<T, 64>(p, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" int lwdetr_attention(const lwdetr_attn_desc* desc, int dtype, void* hip_stream) {
    if (!desc) return LWDETR_ERR_BAD_ARG;
    const lwdetr_attn_desc& p = *desc;
    if (!p.Q || !p.K || !p.VT || !p.out || p.B <= 0 || p.heads <= 0 || p.Tp <= 0 || p.Tp % 4 != 0) return LWDETR_ERR_BAD_ARG;
    if (p.seqs_per_img <= 0 || p.keys_per_seq < 4 || p.keys_per_seq % 4 != 0 || p.sub_stride <= 0 || p.sub_len <= 0 || p.sub_len > p.sub_stride)
        return LWDETR_ERR_BAD_ARG;
    if ((long)(p.seqs_per_img - 1) * p.seq_tok_stride + p.keys_per_seq > p.Tp) return LWDETR_ERR_BAD_ARG;
    if (p.seq_tok_stride % 4 != 0 || p.ldo % 4 != 0) return LWDETR_ERR_BAD_ARG;
    if ((long)p.B * p.seqs_per_img > 65535 || p.heads > 65535) return LWDETR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F32: return dispatch_hd<float>(p, st);
        case DT_F16: return dispatch_hd<f16>(p, st);
        case DT_BF16: return dispatch_hd<bf16>(p, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}
