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
#include <cstdlib>

namespace {

// cross-row reductions on the VALU (v_permlane16_swap / v_permlane32_swap) instead of ds_bpermute through the LDS
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// This file is compiled with -fno-honor-nans (Makefile): otherwise hipcc puts a v_max_f32 x, x canonicalisation in front of
// every fmaxf operand that comes out of an MFMA (it cannot prove the value is not a signalling NaN) - ~15 instructions per
// 8 scores instead of 4 v_max3. Plain VALU does not overlap the matrix pipe on this chip (tools/ubench/issue.hip), so
// softmax instruction count is kernel time. Infinities stay honoured (the key mask is -inf).
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// Tuning builds (tools/attn_ablate.py, -DLWDETR_ATTN_ABL=n, wrong results): 1 = no exp2, 2 = no row max / rescale check,
// 4 = no score MFMAs, 8 = no P V / row-sum MFMAs, 16 = no float -> T conversion of P. Bits combine.
#ifndef LWDETR_ATTN_ABL
#define LWDETR_ATTN_ABL 0
#endif

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

    // Accumulators: O^T tiles, and the softmax denominator as one extra MFMA against an all-ones A operand (every
    // register of lsum[t] then holds the full row sum of query l15: no VALU adds, no cross-lane reduction).
    f32x4 o[QT][DT], lsum[QT];
    // exponent reference of each query (log2 domain), kept negated and replicated: it is the C operand of every score
    // MFMA as it stands (no per-tile broadcast moves), and only changes in the rare rescale path
    f32x4 negm[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        negm[t] = f32x4{0.f, 0.f, 0.f, 0.f}; lsum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    V8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = from_f32<T>(1.f);

    const bool holes = p.sub_len < p.sub_stride;
    constexpr float RESCALE_THR = 8.f;      // lazy rescaling: keep the old reference while no score exceeds it by 2^8

    // K / V^T operand fragments of one 32-key step; loaded one step ahead (software pipelining: the loads of step i+1
    // are in flight while step i runs its MFMAs and exps)
    struct KV { V8 k8[2][NC]; V4 k4[2]; V8 v[DT]; };
    // Key order inside a 32-key step: score tile kt, row 4g + r  <->  key k0 + 8g + 4kt + r. Any row order is as good as
    // another for S^T = K Q^T, and with this one the 8 k-slots a lane owns in O^T = V^T P^T (slots 0-3 from tile 0, 4-7 from
    // tile 1, rows 4g..4g+3 of each) are the 8 CONSECUTIVE keys k0 + 8g .. + 7: one 16-byte V^T load per lane and step.
    auto load_k = [&](int k0, KV& f) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            int key = k0 + 8 * (l15 >> 2) + 4 * kt + (l15 & 3); key = key < last ? key : last;
            if (HD >= 32) {
#pragma unroll
                for (int c = 0; c < NC; ++c) f.k8[kt][c] = *(const V8*)(Kb + (long)key * HD + c * 32 + g * 8);
            } else {
                f.k4[kt] = *(const V4*)(Kb + (long)key * HD + g * 4);
            }
        }
    };
    auto load_v = [&](int k0, KV& f) {
        int kv = k0 + g * 8;
        kv = kv + 7 < nkeys ? kv : (nkeys - 8);                      // whole run in range or clamped (masked below)
        // clamped only in a ragged last step; keys_per_seq is a multiple of 4, so the run then starts 4 (mod 8) keys early
        // (or >= 8 early: every slot is a key past the end and P = 0 there)
        const bool shift4 = k0 + g * 8 - kv == 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            V8 v = *(const V8*)(Vb + (long)(dt * 16 + l15) * p.Tp + kv);
            if (shift4) {                                            // keep slot e <-> key k0 + 8g + e for the 4 valid keys
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[4 + e];
            }
            f.v[dt] = v;
        }
    };
    auto load_kv = [&](int k0, KV& f) { load_k(k0, f); load_v(k0, f); };

    // score tiles of one query tile against the two key tiles of a step (C operand = -reference)
    auto qk_tile = [&](const KV& f, int t, f32x4 (&st)[2]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x4 a = negm[t];
            if (LWDETR_ATTN_ABL & 4) { a[0] += to_f32<T>(f.k4[kt][0]); st[kt] = a; continue; }
            if (HD >= 32) {
#pragma unroll
                for (int c = 0; c < NC; ++c) a = Mma<T>::k32(f.k8[kt][c], q8[t][c], a);
            } else {
                a = Mma<T>::k16(f.k4[kt], q4[t], a);
            }
            st[kt] = a;
        }
    };
    // mask keys beyond the sequence / pad rows inside it, row maxima, lazy rescale (the reference only moves when some
    // score exceeds it by more than RESCALE_THR - always on the first step; O and the denominator follow exactly once)
    auto softmax_head = [&](f32x4 (&s)[QT][2], int k0) {
        // plain selects: a nested-branch formulation of this block was miscompiled by hipcc 7.2 (inner select dropped)
        if (k0 + 32 > nkeys || holes) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + g * 8 + kt * 4 + r;
                    const int sub = holes ? key % p.sub_stride : 0;
                    const bool ok = (key < nkeys) & (sub < p.sub_len);
#pragma unroll
                    for (int t = 0; t < QT; ++t) s[t][kt][r] = ok ? s[t][kt][r] : -INFINITY;
                }
        }
        float lmax[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            lmax[t] = max3(s[t][0][0], s[t][0][1], s[t][0][2]);
            lmax[t] = max3(lmax[t], s[t][0][3], s[t][1][0]);
            lmax[t] = max3(lmax[t], s[t][1][1], s[t][1][2]);
            lmax[t] = max3(lmax[t], s[t][1][3], s[t][1][3]);
        }
        if (LWDETR_ATTN_ABL & 2) { for (int t = 0; t < QT; ++t) lmax[t] = s[t][0][0]; }
        float lall = lmax[0];
#pragma unroll
        for (int t = 1; t + 1 < QT; t += 2) lall = max3(lall, lmax[t], lmax[t + 1]);
        if (QT % 2 == 0) lall = max3(lall, lmax[QT - 1], lmax[QT - 1]);
        const bool need = k0 == 0 || lall > RESCALE_THR;
        if (__any(need)) {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float mx = xor32_max(xor16_max(lmax[t]));           // row max relative to the current reference
                mx = k0 == 0 ? mx : fmaxf(mx, 0.f);                 // the reference never decreases after step 0
                const float alpha = __builtin_amdgcn_exp2f(-mx);
                negm[t] -= mx;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) s[t][kt] -= mx;
                lsum[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[t][dt] *= alpha;
            }
        }
    };
    // P = 2^s of one query tile (already the B operand), row sum and O^T accumulation
    auto pv_tile = [&](const KV& f, int t, const f32x4 (&st)[2]) {
        V8 pf;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (LWDETR_ATTN_ABL & 1) ? st[kt][r] : __builtin_amdgcn_exp2f(st[kt][r]);
                pf[kt * 4 + r] = from_f32<T>(e);
            }
        if (LWDETR_ATTN_ABL & 8) { lsum[t][0] += to_f32<T>(pf[0]) + to_f32<T>(pf[5]); return; }
        lsum[t] = Mma<T>::k32(ones, pf, lsum[t]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[t][dt] = Mma<T>::k32(f.v[dt], pf, o[t][dt]);
    };
    auto step = [&](const KV& f, int k0) {
        f32x4 s[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) qk_tile(f, t, s[t]);
        softmax_head(s, k0);
#pragma unroll
        for (int t = 0; t < QT; ++t) pv_tile(f, t, s[t]);
    };
    // (Tried and measured slower, 185 -> 192 us on the 1600-key shape: issuing the score MFMAs of step i + 1 tile by tile
    // between the exponentials of step i. Transcendentals, MFMAs and plain VALU of the waves of a SIMD add up on this chip
    // whatever the interleaving; only the instruction count moves the time.)

    // The loads of step i + 1 are issued UNCONDITIONALLY (addresses are clamped, a step past the end fetches rows that are
    // never used): with the issue under an `if`, hipcc merges the "nothing new in flight" path into the waits and drains
    // the queue (s_waitcnt vmcnt(0)) right before the MFMAs that need step i's operands - i.e. it waits for the loads it
    // has just issued, one exposed L2 round trip per step.
    // (Short sequences - a 100-key window is 4 steps - keep the conditional form: one wasted fetch per wave costs more
    // there than the drained queue.)
    KV fa, fb;
    if (nkeys >= 512) {
        load_kv(0, fa);
        for (int k0 = 0;;) {
            load_kv(k0 + 32, fb);
            step(fa, k0);
            k0 += 32;
            if (k0 >= nkeys) break;
            load_kv(k0 + 32, fa);
            step(fb, k0);
            k0 += 32;
            if (k0 >= nkeys) break;
        }
    } else {
        load_kv(0, fa);
        for (int k0 = 0;;) {
            if (k0 + 32 < nkeys) load_kv(k0 + 32, fb);
            step(fa, k0);
            k0 += 32;
            if (k0 >= nkeys) break;
            if (k0 + 32 < nkeys) load_kv(k0 + 32, fa);
            step(fb, k0);
            k0 += 32;
            if (k0 >= nkeys) break;
        }
    }

    // ---- normalise and store: lane holds channels dt*16 + 4g .. +3 of query q0 + t*16 + l15
    T* __restrict__ out = (T*)p.out;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float l = lsum[t][0];
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

template <typename T, int HD, int QT>
int launch_qt(const lwdetr_attn_desc& p, hipStream_t st) {
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

template <typename T, int HD>
int launch(const lwdetr_attn_desc& p, hipStream_t st) {
    // long sequences: 64 queries per wave (K / V^T fragments amortised over 4 query tiles); short ones keep 32 so a
    // 100-token window still spreads over 4 waves
    static const char* force = getenv("LWDETR_ATTN_QT");
    // ... and only when that still leaves at least one workgroup per CU (a single image has 84 of them at 64 queries per
    // wave: 25 us per launch against 18 us with 32)
    const long wgs4 = (long)((p.keys_per_seq + 255) / 256) * p.heads * p.B * p.seqs_per_img;
    const bool qt4 = force ? atoi(force) == 4 : (p.keys_per_seq >= 1024 && HD <= 32 && sizeof(T) == 2 && wgs4 >= 256);
    if (qt4) return launch_qt<T, HD, (HD <= 32 ? 4 : 2)>(p, st);
    return launch_qt<T, HD, 2>(p, st);
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
    if (p.seqs_per_img <= 0 || p.keys_per_seq < 8 || p.keys_per_seq % 4 != 0 || p.sub_stride <= 0 || p.sub_len <= 0 || p.sub_len > p.sub_stride)
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
