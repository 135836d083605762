// Fused ViT block tail for gfx950, 16-bit types, C in {192, 384} (round 3; replaces mlp_kernel at BASELINE batch sizes):
//
//   x1 = x + gamma1 * (att Wp^T + bp)                       attention output projection   (vit.py:138, :206-216)
//   x  = x1 + gamma2 * (fc2(GELU(fc1(LN2(x1)))) + b2)       MLP                           (vit.py:217-218, timm Mlp)
//   q, k, v^T = heads(LN1'(x) Wqkv'^T + b')                 norm1 + QKV of the NEXT block (vit.py:199, :123-130), optional
//
// Why it looks the way it does (measured on the round-2 kernel, profiles/r2*):
//   * the matrix pipe, the VALU (GELU: 2 transcendentals + 7 plain instructions per hidden value) and the LDS -> register
//     fragment returns all cost about the same per hidden chunk at C = 192, and the old kernel ran them one after the other
//     (all 8 waves in lockstep: MFMA phase, GELU phase, barrier). Here ONE wave per SIMD (4 per workgroup, 512 registers)
//     owns 64 tokens (C = 192) / 32 tokens (C = 384) and runs a software pipeline: while GELU of chunk k runs on the VALU,
//     the same instruction stream issues fc2 of chunk k-1 and fc1 of chunk k+1 (one 32x32x16 MFMA every ~7 VALU
//     instructions, pinned with sched_barrier). Twice the tokens per wave = half the weight-fragment traffic per MFMA.
//   * every weight of the block (Wp, W1, W2, Wqkv') is packed ON THE HOST into one stream of 32-row pieces, each piece a
//     sequence of 1 KB MFMA fragments in lane order (lane l's 16 bytes at l * 16): the DMA is linear, a fragment read is
//     `base + lane * 16 + immediate`, conflict-free by construction, no swizzle arithmetic, no padded rows. The stream
//     runs through ONE ring of NSLOT pieces with counted s_waitcnt vmcnt + raw s_barrier per step (2 pieces), several
//     steps ahead - the old kernel restarted a 2-deep double buffer (and paid a DMA round trip) per phase.
//   * the residual stream never sits in LDS or goes back to HBM between the projection and the epilogue: the fc2
//     accumulators start at x1 / gamma2 + b2 and the epilogue is one multiply (|x1 / gamma2| only costs f32 round-off of
//     x1, far below the 16-bit output grid; the host refuses gamma2 entries that are zero / denormal).
// Fragment conventions (32x32x16 MFMA, wave64): A lane (i = l & 31, h = l >> 5) holds A[i][8h .. 8h+7]; B the same with
// j = l & 31; D register 4 b + e of lane (j, h) is D[8 b + 4 h + e][j]. An accumulator tile therefore IS the B operand
// of the next contraction over its rows, with k-slot (step 2n + beta, half h, s = 4 b' + e) <-> row 32 n + 16 beta + 8 b' +
// 4 h + e: the host permutes the weight columns accordingly (lwdetr_amd/kernels.py:pack_vit_block).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

template <typename T> struct Mma32;
template <> struct Mma32<f16> {
    static __device__ __forceinline__ f32x16 k16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<bf16> {
    static __device__ __forceinline__ f32x16 k16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <typename T> struct Pk;
template <> struct Pk<f16> { typedef f16x2 v2; };
template <> struct Pk<bf16> { typedef bf16x2 v2; };
// two f32 -> one packed dword of T (round to nearest even: v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32)
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, typename Pk<T>::v2));
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct VbParams {
    void* x; long ldx;                 // (M, C) residual stream, updated in place
    const void* att; long ldatt;       // (M, C) attention output (heads concatenated)
    const void* wstream;               // packed weight pieces (see pack_vit_block)
    const float* vec;                  // packed f32 vectors: b1'[4C] | bp[C] | g1[C] | 1/g1[C] | b2[C] | 1/g2[C] | g2[C] | bqkv'[3C], padded
    void* out2; long ld2;              // optional copy of the new rows (ViT feature taps)
    float* stats_out;                  // optional (M, 2): mean, rstd of the new rows (eps_next)
    void* q; void* k; void* vt;        // QKV outputs of the next block
    long M;
    float eps, eps_next, qscale;
    int heads, hd_log2, Tp;
    unsigned qkv_bytes;                // size of each of q / k / vt in bytes (buffer bound)
};

#define VB_VMW(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// wait until at most n vector-memory operations of this wave are outstanding (n wave-uniform, rounded down to a multiple of 3)
__device__ __forceinline__ void vb_wait_le(int n) {
    if (n >= 63) { VB_VMW(63); return; }
    switch (n / 3) {
        case 0: VB_VMW(0); break;   case 1: VB_VMW(3); break;   case 2: VB_VMW(6); break;   case 3: VB_VMW(9); break;
        case 4: VB_VMW(12); break;  case 5: VB_VMW(15); break;  case 6: VB_VMW(18); break;  case 7: VB_VMW(21); break;
        case 8: VB_VMW(24); break;  case 9: VB_VMW(27); break;  case 10: VB_VMW(30); break; case 11: VB_VMW(33); break;
        case 12: VB_VMW(36); break; case 13: VB_VMW(39); break; case 14: VB_VMW(42); break; case 15: VB_VMW(45); break;
        case 16: VB_VMW(48); break; case 17: VB_VMW(51); break; case 18: VB_VMW(54); break; case 19: VB_VMW(57); break;
        default: VB_VMW(60); break;
    }
}

// GELU for 16-bit storage (common.h:gelu_fast16) split into three stages of 3 VALU-class instructions per value, so that the
// hidden loop can hand them out between MFMAs: s0 -> (x2, p), s1 -> e = exp2(x * (p x2 + c0)), s2 -> x * rcp(1 + e).
__device__ __forceinline__ void gelu_s0(float x, float& x2, float& p) {
    x2 = fminf(x * x, 36.f);
    p = fmaf(x2, 0.0010142630555f, -0.1067757240036f);
}
__device__ __forceinline__ float gelu_s1(float x, float x2, float p) {
    return __builtin_amdgcn_exp2f(x * fmaf(p, x2, -2.3011213394584f));
}
__device__ __forceinline__ float gelu_s2(float x, float e) { return x * __builtin_amdgcn_rcpf(1.f + e); }

template <typename T, int C, int NH, bool QKV>
__global__ __launch_bounds__(256, 1) void vitblock_kernel(const VbParams p) {
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    static_assert(sizeof(T) == 2, "16-bit types only");
    constexpr int KS = C / 16;                  // k-steps of a K = C contraction = fragments per piece
    constexpr int NTI = C / 32;                 // 32-row tiles along C
    constexpr int NCH = C / 8;                  // hidden chunks of 32 units (4C / 32)
    constexpr int PIECE_B = KS * 1024;          // bytes per piece
    constexpr int DPW = KS / 4;                 // DMA wave-instructions per piece and wave
    constexpr int NSLOT = C == 192 ? 8 : 5;     // ring depth in pieces
    constexpr int VEC_F = 13 * C;
    constexpr int VEC_B = (VEC_F * 4 + 4095) / 4096 * 4096, VEC_DPW = VEC_B / 4096;
    constexpr int NP_PROJ = NTI, NP_HID = 2 * NCH, NP_QKV = QKV ? 3 * NTI : 0, NP = NP_PROJ + NP_HID + NP_QKV;
    constexpr int H0 = NP_PROJ, Q0 = NP_PROJ + NP_HID;
    constexpr int RD = NH == 2 ? 3 : 5;         // fragment read-ahead (fragments)
    static_assert(DPW % 3 == 0 && VEC_DPW >= 1, "wait counts are kept in multiples of 3");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float* vec = (const float*)(smem + NSLOT * PIECE_B);
    const float* b1s = vec; const float* bps = vec + 4 * C; const float* g1s = bps + C; const float* rg1s = g1s + C;
    const float* b2s = rg1s + C; const float* rg2s = b2s + C; const float* g2s = rg2s + C; const float* bqs = g2s + C;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;

    // ---- this wave's tokens: 4-token units dealt evenly over all waves of the grid (at most 32 * NH per wave: host)
    const long U = p.M >> 2, nwv = (long)gridDim.x * 4, wg = (long)blockIdx.x * 4 + wave;
    const long u0 = wg * U / nwv, u1 = (wg + 1) * U / nwv;
    const long t0 = u0 * 4;
    const int nvalid = (int)(u1 - u0) * 4;

    // ---- weight stream: linear LDS-DMA, 1 KB per wave-instruction (inline asm: hipcc must not turn the pending pieces into
    // lgkmcnt(0) drains of the fragment reads, cf. mlp.hip). Piece i lives in ring slot i % NSLOT.
    const char* wsrc = (const char*)p.wstream;
    auto dma1k = [&](const char* src_uniform, unsigned voff, unsigned lds_dst) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)src_uniform);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)src_uniform >> 32));
        const char* sp = (const char*)(((uintptr_t)hi << 32) | lo);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(sp) : "memory");
    };
    auto dma_piece = [&](int piece) {
        const unsigned slot = (unsigned)piece % NSLOT;
        const char* src = wsrc + (size_t)piece * PIECE_B;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const unsigned kb = (unsigned)(wave * DPW + i) * 1024u;
            dma1k(src, kb + lane16, lds0 + slot * PIECE_B + kb);
        }
    };
    int issued = 0;
    // step boundary: the step consumes pieces [a, b), everything below a is dead. `extra` = vector-memory operations this wave
    // is KNOWN to have issued after its DMA of piece b - 1 besides later pieces (a lower bound is safe, it only waits longer).
    auto boundary = [&](int a, int b, int extra) {
        vb_wait_le((issued - b) * DPW + extra);
        __builtin_amdgcn_s_barrier();
        int lim = a + NSLOT; lim = lim < NP ? lim : NP;
        while (issued < lim) { dma_piece(issued); ++issued; }
    };
    auto frag = [&](int piece, int f) -> V8 {
        return *(const V8*)(smem + ((unsigned)piece % NSLOT) * PIECE_B + f * 1024 + lane16);
    };

    // ---- prologue: vectors + the first NSLOT pieces in flight, then the attention rows as B fragments
    {
        const char* vsrc = (const char*)p.vec;
#pragma unroll
        for (int i = 0; i < VEC_DPW; ++i) {
            const unsigned kb = (unsigned)(wave * VEC_DPW + i) * 1024u;
            dma1k(vsrc, kb + lane16, lds0 + NSLOT * PIECE_B + kb);
        }
        for (; issued < NSLOT; ++issued) dma_piece(issued);
    }
    const T* att_w = (const T*)p.att + t0 * p.ldatt;
    T* x_w = (T*)p.x + t0 * p.ldx;
    const __amdgpu_buffer_rsrc_t r_att = __builtin_amdgcn_make_buffer_rsrc((void*)att_w, 0, (int)(nvalid * p.ldatt * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)x_w, 0, (int)(nvalid * p.ldx * 2), 0x00020000);
    V8 xf[NH][KS];                              // attention rows, later LN(x1), later LN'(x): B operands, k-run of 8 per k-step
#pragma unroll
    for (int th = 0; th < NH; ++th)
#pragma unroll
        for (int t = 0; t < KS; ++t)
            xf[th][t] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(
                r_att, (unsigned)(((32 * th + j) * p.ldatt + 16 * t + 8 * h) * 2), 0, 0));

    f32x16 acc2[NTI][NH];
    // ---- attention output projection: D[channel][token], Wp pieces of 32 output channels, 2 pieces per step. The accumulators
    // start at x / gamma1 + bp (x in accumulator layout: 4 channels 32 n + 8 b + 4 h .. of token j), so x1 = gamma1 * acc and the
    // x rows are consumed before the first MFMA (no second copy of the residual stream in registers).
    {
        V4 xv[NH][NTI][4];
#pragma unroll
        for (int th = 0; th < NH; ++th)
#pragma unroll
            for (int n = 0; n < NTI; ++n)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    xv[th][n][b] = __builtin_bit_cast(V4, __builtin_amdgcn_raw_buffer_load_b64(
                        r_x, (unsigned)(((32 * th + j) * p.ldx + 32 * n + 8 * b + 4 * h) * 2), 0, 0));
        // loads issued after the DMA of the initial pieces: the attention and x rows
        boundary(0, 2, NH * KS + NH * NTI * 4);
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            f32x16 na[NH];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c0 = 32 * n + 8 * b + 4 * h;
                const f32x4 bb = *(const f32x4*)(bps + c0), rg = *(const f32x4*)(rg1s + c0);
#pragma unroll
                for (int th = 0; th < NH; ++th) {
                    const f32x4 xo = up4<T>(xv[th][n][b]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) na[th][4 * b + e] = fmaf(xo[e], rg[e], bb[e]);
                }
            }
#pragma unroll
            for (int th = 0; th < NH; ++th) {
                asm volatile("" : "+a"(na[th]));      // complete tile, in the accumulator file (see the LayerNorm section)
                acc2[n][th] = na[th];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int s = 0; s < NTI / 2; ++s) {
        if (s > 0) boundary(2 * s, 2 * s + 2, 2 * s + 1 < NSLOT ? NH * KS + NH * NTI * 4 : 0);
        {   // 2 KS fragments through the read-ahead ring, one read per fragment retired (hipcc would hoist all 2 KS reads)
            V8 fr[RD];
#pragma unroll
            for (int i = 0; i < RD; ++i) fr[i] = frag(2 * s + i / KS, i % KS);
#pragma unroll
            for (int fi = 0; fi < 2 * KS; ++fi) {
                const int n = 2 * s + fi / KS, t = fi % KS;
                const V8 a = fr[fi % RD];
#pragma unroll
                for (int th = 0; th < NH; ++th) acc2[n][th] = Mma32<T>::k16(a, xf[th][t], acc2[n][th]);
                if (fi + RD < 2 * KS) fr[fi % RD] = frag(2 * s + (fi + RD) / KS, (fi + RD) % KS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- x1 = gamma1 * acc rounded to the storage type; LayerNorm (affine folded into W1 / b1) -> B fragments; the fc2
    // accumulators start at x1 / gamma2 + b2 (the epilogue is out = gamma2 * acc). x1 is held as packed 16-bit pairs between
    // the three passes (sum, variance, normalise): 48 registers per token half instead of 96 f32 values.
    auto unpack2 = [](unsigned w, float& a, float& b) {
        const typename Pk<T>::v2 v = __builtin_bit_cast(typename Pk<T>::v2, w);
        a = to_f32<T>(v[0]); b = to_f32<T>(v[1]);
    };
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        unsigned xp[NTI][8];                    // dword d of tile n: registers 2 d, 2 d + 1 of the accumulator tile
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 gg = *(const f32x4*)(g1s + 32 * n + 8 * b + 4 * h);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned w = pack2<T>(gg[2 * d] * acc2[n][th][4 * b + 2 * d], gg[2 * d + 1] * acc2[n][th][4 * b + 2 * d + 1]);
                    xp[n][2 * b + d] = w;
                    float v0, v1; unpack2(w, v0, v1);
                    s += v0 + v1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        s += __shfl_xor(s, 32);
        const float mean = s * (1.f / C);
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n)
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                float v0, v1; unpack2(xp[n][d], v0, v1);
                v0 -= mean; v1 -= mean;
                v = fmaf(v0, v0, v); v = fmaf(v1, v1, v);
            }
        __builtin_amdgcn_sched_barrier(0);
        v += __shfl_xor(v, 32);
        const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps);
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            u32x4 w0, w1;
            f32x16 na;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c0 = 32 * n + 8 * b + 4 * h;
                const f32x4 rg = *(const f32x4*)(rg2s + c0), b2 = *(const f32x4*)(b2s + c0);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    float v0, v1; unpack2(xp[n][2 * b + d], v0, v1);
                    na[4 * b + 2 * d] = fmaf(v0, rg[2 * d], b2[2 * d]);
                    na[4 * b + 2 * d + 1] = fmaf(v1, rg[2 * d + 1], b2[2 * d + 1]);
                    const unsigned nw = pack2<T>((v0 - mean) * rstd, (v1 - mean) * rstd);
                    if (b < 2) w0[2 * b + d] = nw; else w1[2 * (b - 2) + d] = nw;
                }
            }
            asm volatile("" : "+a"(na));            // the new accumulator tile is complete (and in the accumulator file) here
            acc2[n][th] = na;
            xf[th][2 * n] = __builtin_bit_cast(V8, w0);
            xf[th][2 * n + 1] = __builtin_bit_cast(V8, w1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- hidden loop, software pipelined. Pieces after the projection: W1c(0), W1c(1), then (W2c(k-1), W1c(k+1)) for k = 1 ..
    // NCH-2, then W2c(NCH-2), W2c(NCH-1). Iteration k: GELU(k) on the VALU, fc2(k-1) and fc1(k+1) on the matrix pipe.
    f32x16 acc1[2][NH];
    u32x4 hf[2][NH][2];                         // GELU output as B operands (packed pairs): [buffer][token half][k-step of the chunk]
    auto bias16 = [&](const float* src) -> f32x16 {      // src[8 b + 4 h + e] -> register 4 b + e
        f32x16 r;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 v = *(const f32x4*)(src + 8 * b + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[4 * b + e] = v[e];
        }
        return r;
    };
    // one pipelined iteration. CUR: acc1 / hf buffer of chunk k; p2 / p1: pieces W2c(k-1) / W1c(k+1) (ignored when the half is off)
    auto iter = [&](auto cur_tag, auto fc2_tag, auto fc1_tag, int p2, int p1, int k) {
        constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
        constexpr bool DO2 = decltype(fc2_tag)::value, DO1 = decltype(fc1_tag)::value;
        constexpr int NF2 = DO2 ? 2 * NTI : 0, NF1 = DO1 ? KS : 0, NF = NF2 + NF1, S = NF * NH;
        constexpr int TK = 24 * NH;                           // GELU ticks: 8 NH value pairs x 3 stages
        auto fragi = [&](int i) -> V8 { return i < NF2 ? frag(p2, i) : frag(p1, i - NF2); };
        f32x16 bias = {};
        V8 fr[RD];
#pragma unroll
        for (int i = 0; i < RD; ++i) if (i < NF) fr[i] = fragi(i);
        float gx2[2], gp[2];
#pragma unroll
        for (int m = 0; m < S; ++m) {
            // GELU ticks of this slot
#pragma unroll
            for (int ti = m * TK / S; ti < (m + 1) * TK / S; ++ti) {
                const int pair = ti / 3, st = ti % 3, th = pair / 8, r0 = (pair % 8) * 2;
                if (st == 0) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) gelu_s0(acc1[CUR][th][r0 + u], gx2[u], gp[u]);
                    asm volatile("" : "+v"(gx2[0]), "+v"(gx2[1]), "+v"(gp[0]), "+v"(gp[1]));
                } else if (st == 1) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) gp[u] = gelu_s1(acc1[CUR][th][r0 + u], gx2[u], gp[u]);
                    asm volatile("" : "+v"(gp[0]), "+v"(gp[1]));
                } else {
                    const int bq = r0 >> 2, e0 = r0 & 3;             // register 4 bq + e0: k-step bq / 2 of the chunk, dword 2 (bq & 1) + e0 / 2
                    unsigned w = pack2<T>(gelu_s2(acc1[CUR][th][r0], gp[0]), gelu_s2(acc1[CUR][th][r0 + 1], gp[1]));
                    asm volatile("" : "+v"(w));
                    hf[CUR][th][bq >> 1][2 * (bq & 1) + (e0 >> 1)] = w;
                }
            }
            const int fi = m / NH, th = m % NH;
            if (DO1 && m == NF2 * NH - (NF2 ? 4 : 0)) bias = bias16(b1s + (k + 1) * 32);     // short live range: just ahead of fc1
            const V8 a = fr[fi % RD];
            if (fi < NF2) {
                const int kap = fi / NTI, n = fi % NTI;
                acc2[n][th] = Mma32<T>::k16(a, __builtin_bit_cast(V8, hf[NXT][th][kap]), acc2[n][th]);
            } else {
                const int t = fi - NF2;
                acc1[NXT][th] = Mma32<T>::k16(a, xf[th][t], t == 0 ? bias : acc1[NXT][th]);
            }
            if (th == NH - 1 && fi + RD < NF) fr[fi % RD] = fragi(fi + RD);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;
    {   // pre-step: fc1(0)
        boundary(H0, H0 + 1, 0);
        const f32x16 bias = bias16(b1s);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const V8 a = frag(H0, t);
#pragma unroll
            for (int th = 0; th < NH; ++th) acc1[0][th] = Mma32<T>::k16(a, xf[th][t], t == 0 ? bias : acc1[0][th]);
        }
    }
    boundary(H0 + 1, H0 + 2, 0);
    iter(I0{}, No{}, Yes{}, 0, H0 + 1, 0);
#pragma unroll 1
    for (int k = 1; k < NCH - 1; k += 2) {
        boundary(H0 + 2 * k, H0 + 2 * k + 2, 0);
        iter(I1{}, Yes{}, Yes{}, H0 + 2 * k, H0 + 2 * k + 1, k);
        boundary(H0 + 2 * k + 2, H0 + 2 * k + 4, 0);
        iter(I0{}, Yes{}, Yes{}, H0 + 2 * k + 2, H0 + 2 * k + 3, k + 1);
    }
    boundary(H0 + 2 * NCH - 2, H0 + 2 * NCH - 1, 0);
    iter(I1{}, Yes{}, No{}, H0 + 2 * NCH - 2, 0, NCH - 1);
    {   // post-step: fc2(NCH - 1)
        boundary(H0 + 2 * NCH - 1, H0 + 2 * NCH, 0);
#pragma unroll
        for (int fi = 0; fi < 2 * NTI; ++fi) {
            const V8 a = frag(H0 + 2 * NCH - 1, fi);
            const int kap = fi / NTI, n = fi % NTI;
#pragma unroll
            for (int th = 0; th < NH; ++th) acc2[n][th] = Mma32<T>::k16(a, __builtin_bit_cast(V8, hf[1][th][kap]), acc2[n][th]);
        }
    }

    // ---- epilogue: out = gamma2 * acc (rounded), stores, statistics of the new rows, LN'(x) as the next B operand
    const __amdgpu_buffer_rsrc_t r_o2 = __builtin_amdgcn_make_buffer_rsrc(
        p.out2 ? (void*)((T*)p.out2 + t0 * p.ld2) : (void*)x_w, 0, p.out2 ? (int)(nvalid * p.ld2 * 2) : 0, 0x00020000);
    const bool has_o2 = p.out2 != nullptr;
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c0 = 32 * n + 8 * b + 4 * h;
                const f32x4 gg = *(const f32x4*)(g2s + c0);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = gg[e] * acc2[n][th][4 * b + e];
                const V4 ov = cvt4<T>(o);
                const u32x2 ow = __builtin_bit_cast(u32x2, ov);
                __builtin_amdgcn_raw_buffer_store_b64(ow, r_x, (unsigned)(((32 * th + j) * p.ldx + c0) * 2), 0, 0);
                if (has_o2) __builtin_amdgcn_raw_buffer_store_b64(ow, r_o2, (unsigned)(((32 * th + j) * p.ld2 + c0) * 2), 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float v = to_f32<T>(ov[e]); acc2[n][th][4 * b + e] = v; s += v; }
                if (b == 3) __builtin_amdgcn_sched_barrier(0);
            }
        if (p.stats_out || QKV) {
            s += __shfl_xor(s, 32);
            const float mean = s * (1.f / C);
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < NTI; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float dl = acc2[n][th][e] - mean; v += dl * dl; }
            v += __shfl_xor(v, 32);
            const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps_next);
            if (p.stats_out && h == 0 && 32 * th + j < nvalid) {
                float* so = p.stats_out + 2 * (t0 + 32 * th + j);
                so[0] = mean; so[1] = rstd;
            }
            if (QKV) {
#pragma unroll
                for (int n = 0; n < NTI; ++n)
#pragma unroll
                    for (int be = 0; be < 2; ++be)
                        {
                            u32x4 w;
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const int r = (2 * be + (d >> 1)) * 4 + (d & 1) * 2;
                                w[d] = pack2<T>((acc2[n][th][r] - mean) * rstd, (acc2[n][th][r + 1] - mean) * rstd);
                            }
                            xf[th][2 * n + be] = __builtin_bit_cast(V8, w);
                        }
            }
        }
    }

    if (QKV) {
        // ---- chained norm1 + QKV of the next block: pieces of 32 features (q: 0 .. NTI-1, k: NTI .. 2 NTI-1, v: 2 NTI ..).
        // Q, K: D[feature][token] -> (B, heads, Tp, hd); V: operands swapped, D[token][feature] -> V^T (B, heads, hd, Tp).
        const __amdgpu_buffer_rsrc_t r_q = __builtin_amdgcn_make_buffer_rsrc(p.q, 0, (int)p.qkv_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_k = __builtin_amdgcn_make_buffer_rsrc(p.k, 0, (int)p.qkv_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc(p.vt, 0, (int)p.qkv_bytes, 0x00020000);
        const int hd = 1 << p.hd_log2;
        unsigned row_qk[NH], row_v[NH][4];        // element offsets of this lane's token (q, k) / 4-token runs (v^T); ~0u = no token
#pragma unroll
        for (int th = 0; th < NH; ++th) {
            const unsigned tok = (unsigned)t0 + 32 * th + j;
            const unsigned img = tok / (unsigned)p.Tp, wi = tok - img * (unsigned)p.Tp;
            row_qk[th] = 32 * th + j < nvalid ? (unsigned)(((long)img * p.heads * p.Tp + wi) << p.hd_log2) : 0x7fffffffu;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int tl = 32 * th + 8 * b + 4 * h;
                const unsigned tk = (unsigned)t0 + tl;
                const unsigned im = tk / (unsigned)p.Tp, wv = tk - im * (unsigned)p.Tp;
                row_v[th][b] = tl < nvalid ? (unsigned)((long)im * p.heads * hd * p.Tp + wv) : 0x7fffffffu;
            }
        }
        constexpr int XST = NH * NTI * 4;         // epilogue stores per wave (x); the tap copy doubles it
        const int est = XST * (has_o2 ? 2 : 1);
#pragma unroll 1
        for (int s = 0; s < NP_QKV / 2; ++s) {
            // stores issued since the DMA of the step's pieces: a lower bound that is exact in the steady state would need the
            // per-step store count; the epilogue stores (all issued after the DMA of the first QKV pieces) are counted
            boundary(Q0 + 2 * s, Q0 + 2 * s + 2, s == 0 ? est : 0);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int pi = 2 * s + pp, piece = Q0 + pi;
                const int sg = pi / NTI, nl0 = (pi - sg * NTI) * 32;
                if (sg < 2) {
                    const f32x16 bias = bias16(bqs + sg * C + nl0);
                    f32x16 acc[NH];
#pragma unroll
                    for (int t = 0; t < KS; ++t) {
                        const V8 a = frag(piece, t);
#pragma unroll
                        for (int th = 0; th < NH; ++th) acc[th] = Mma32<T>::k16(a, xf[th][t], t == 0 ? bias : acc[th]);
                    }
                    const float sc = sg == 0 ? p.qscale : 1.f;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int f = nl0 + 8 * b + 4 * h, hh = f >> p.hd_log2, dd = f & (hd - 1);
                        const unsigned col = (unsigned)(((long)hh * p.Tp << p.hd_log2) + dd);
#pragma unroll
                        for (int th = 0; th < NH; ++th) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[th][4 * b + e] * sc;
                            const u32x2 ow = __builtin_bit_cast(u32x2, cvt4<T>(o));
                            const unsigned off = row_qk[th] == 0x7fffffffu ? 0xfffffff0u : (row_qk[th] + col) * 2u;
                            if (sg == 0) __builtin_amdgcn_raw_buffer_store_b64(ow, r_q, off, 0, 0);
                            else __builtin_amdgcn_raw_buffer_store_b64(ow, r_k, off, 0, 0);
                        }
                    }
                } else {
                    const float bv = bqs[2 * C + nl0 + j];
                    f32x16 binit;
#pragma unroll
                    for (int e = 0; e < 16; ++e) binit[e] = bv;
                    f32x16 acc[NH];
#pragma unroll
                    for (int t = 0; t < KS; ++t) {
                        const V8 a = frag(piece, t);
#pragma unroll
                        for (int th = 0; th < NH; ++th) acc[th] = Mma32<T>::k16(xf[th][t], a, t == 0 ? binit : acc[th]);
                    }
                    const int f = nl0 + j, hh = f >> p.hd_log2, dd = f & (hd - 1);
                    const unsigned rowb = (unsigned)(((long)hh * hd + dd) * p.Tp);
#pragma unroll
                    for (int th = 0; th < NH; ++th)
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            f32x4 o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = acc[th][4 * b + e];
                            const u32x2 ow = __builtin_bit_cast(u32x2, cvt4<T>(o));
                            const unsigned off = row_v[th][b] == 0x7fffffffu ? 0xfffffff0u : (row_v[th][b] + rowb) * 2u;
                            __builtin_amdgcn_raw_buffer_store_b64(ow, r_v, off, 0, 0);
                        }
                }
            }
        }
    }
}

struct VbLaunchState { bool attr_done; int ncu; };

template <typename T, int C, int NH, bool QKV>
int launch_vb(const VbParams& p, hipStream_t st) {
    constexpr int KS = C / 16, PIECE_B = KS * 1024, NSLOT = C == 192 ? 8 : 5;
    constexpr int VEC_B = (13 * C * 4 + 4095) / 4096 * 4096;
    constexpr size_t lds = (size_t)NSLOT * PIECE_B + VEC_B;
    static VbLaunchState state[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    VbLaunchState& s = state[dev];
    if (!s.attr_done) {
        if (hipFuncSetAttribute((const void*)vitblock_kernel<T, C, NH, QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LWDETR_ERR_LAUNCH;
        s.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        s.attr_done = true;
    }
    // grid: whole rounds of one workgroup per CU, tokens dealt evenly (every wave at most 32 NH tokens)
    const long per_wg = 4L * 32 * NH;
    const long need = (p.M + per_wg - 1) / per_wg;
    static const char* env = getenv("LWDETR_VB_GRID");                 // tuning: workgroups (>= need)
    long grid = (need + s.ncu - 1) / s.ncu * s.ncu;
    if (env && atol(env) >= need) grid = atol(env);
    // 4-token units are dealt by floor(): a wave can get one unit more than the average
    while (((p.M / 4 + grid * 4 - 1) / (grid * 4)) * 4 > 32 * NH) ++grid;
    ProfScope ps(KID_MLP, (16.0 + 2.0 + (QKV ? 6.0 : 0.0)) * p.M * C * C,
                 (double)p.M * C * sizeof(T) * 3 + (QKV ? 3.0 : 0.0) * p.M * C * sizeof(T) + (p.out2 ? 1.0 : 0.0) * p.M * C * sizeof(T), st);
    hipLaunchKernelGGL((vitblock_kernel<T, C, NH, QKV>), dim3((unsigned)grid), dim3(256), lds, st, p);
    return lwdetr_check_launch();
}

template <typename T>
int dispatch_vb(const VbParams& p, int C, bool qkv, hipStream_t st) {
    if (C == 192) return qkv ? launch_vb<T, 192, 2, true>(p, st) : launch_vb<T, 192, 2, false>(p, st);
    if (C == 384) return qkv ? launch_vb<T, 384, 1, true>(p, st) : launch_vb<T, 384, 1, false>(p, st);
    return LWDETR_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" long lwdetr_vit_block_stream_bytes(int C, int has_qkv) {
    if (C != 192 && C != 384) return -LWDETR_ERR_UNSUPPORTED;
    const long np = C / 32 + 2 * (C / 8) + (has_qkv ? 3 * (C / 32) : 0);
    return np * (C / 16) * 1024L;
}
extern "C" long lwdetr_vit_block_vec_floats(int C) { return ((13L * C * 4 + 4095) / 4096 * 4096) / 4; }

extern "C" int lwdetr_vit_block(void* x, long ldx, const void* att, long ldatt, const void* wstream, const float* vec, void* out2,
                                long ld2, float* stats_out, long M, int C, float eps, float eps_next, int has_qkv, void* q_out,
                                void* k_out, void* vt_out, float qscale, int heads, int hd, int Tp, int dtype, void* hip_stream) {
    if (!x || !att || !wstream || !vec || M < 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    if (M % 4 != 0 || ldx % 4 != 0 || ldatt % 8 != 0 || (out2 && ld2 % 4 != 0)) return LWDETR_ERR_BAD_ARG;
    if (((uintptr_t)wstream | (uintptr_t)vec | (uintptr_t)att) % 16 != 0 || (uintptr_t)x % 8 != 0) return LWDETR_ERR_BAD_ARG;
    VbParams p = {};
    p.x = x; p.ldx = ldx; p.att = att; p.ldatt = ldatt; p.wstream = wstream; p.vec = vec; p.out2 = out2; p.ld2 = ld2;
    p.stats_out = stats_out; p.M = M; p.eps = eps; p.eps_next = eps_next; p.qscale = qscale;
    if (has_qkv) {
        if (!q_out || !k_out || !vt_out || heads <= 0 || hd < 4 || (hd & (hd - 1)) != 0 || heads * hd != C || Tp <= 0 || Tp % 4 != 0)
            return LWDETR_ERR_BAD_ARG;
        if (M % Tp != 0 || (double)M * C * 2.0 >= 4294967000.0) return LWDETR_ERR_UNSUPPORTED;
        int l2 = 0; while ((1 << l2) < hd) ++l2;
        p.q = q_out; p.k = k_out; p.vt = vt_out; p.heads = heads; p.hd_log2 = l2; p.Tp = Tp;
        p.qkv_bytes = (unsigned)((unsigned long)M * C * 2ul);
    }
    // row offsets inside a wave's tile are 32-bit
    if ((double)(ldx > ldatt ? ldx : ldatt) * 64 * 2 >= 2147483000.0 || (out2 && (double)ld2 * 64 * 2 >= 2147483000.0)) return LWDETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F16: return dispatch_vb<f16>(p, C, has_qkv != 0, st);
        case DT_BF16: return dispatch_vb<bf16>(p, C, has_qkv != 0, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}
