// Fused MFMA GEMM for gfx950:  out = epilogue( A_view(M,K) x W(N,K)^T )
//
// One kernel family serves every dense contraction of the LW-DETR forward path (Linear, 1x1 / 3x3 / stride-2 conv,
// 2x2 transposed conv, the 16x16 patch embedding) - the activations are token-major (NHWC) end to end, so all of them
// are the same "rows x K" problem with a different A-row gather and a different epilogue:
//   A views   : PLAIN (optionally A + A2), CONV3x3 (implicit GEMM, zero padded, stride 1|2, reads raster or window-major
//               token layouts), PATCH16 (gathers 16x16x3 patches of the NCHW image, emits window-major token rows).
//   epilogues : bias, ReLU / erf-GELU / SiLU, scale, LayerScale * x + residual, row masking, second destination (taps),
//               and output layouts LINEAR, HEADS (B,heads,T,hd), HEADS_T (B,heads,hd,T), TOKMAP (row permutation between
//               token layouts, e.g. window-major -> raster, or into a level slice of `memory`), DECONV2x2 (pixel shuffle).
// Tiling: 256 threads = 4 waves (2 x 2), block tile BM x BN x 32, double-buffered LDS (one barrier per k-tile), global
// loads of k-tile t+1 are in flight in registers while tile t feeds the MFMAs. Both operands are K-contiguous, so an
// MFMA fragment is one 16-byte LDS read per lane.  Operand order of the MFMA is chosen per column segment so that each
// lane ends up with 4 consecutive output elements along the destination's contiguous axis (8/16-byte stores):
//   ROW  orientation: mfma(Wfrag, Xfrag) -> lane holds 4 consecutive n for one row m      (all modes but HEADS_T)
//   COL  orientation: mfma(Xfrag, Wfrag) -> lane holds 4 consecutive m for one column n   (HEADS_T: V^T for attention)
// Workgroup ids are remapped so that tiles sharing an A row-panel run on the same XCD (private L2).
#include "common.h"
#include <cstdlib>
#include <type_traits>

// gemm_pt.hip: launches the persistent large-tile kernel if the shape is one of its own (taken = true), otherwise leaves the launch to this file
int lwdetr_gemm_pt_try(const lwdetr_gemm_desc& d, int dtype, hipStream_t st, bool& taken);

namespace {

constexpr int BK = 32;

// ---- epilogue helpers -------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float act_apply(float x, int act) {
    if (act == ACT_GELU) return gelu_for<T>(x);
    if (act == ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
    if (act == ACT_RELU) return x > 0.f ? x : 0.f;
    return x;
}

// destination offset of row m (all modes are separable: off = row_offset(m) + col_offset(n)); returns false for pad rows
__device__ __forceinline__ bool row_offset(const lwdetr_gemm_seg& sg, long m, long& off) {
    if (sg.mode == LWDETR_OUT_LINEAR) { off = m * sg.ldo; return true; }
    if (sg.mode == LWDETR_OUT_HEADS) {
        const int b = (int)(m / sg.p0), t = (int)(m - (long)b * sg.p0);
        off = ((long)b * sg.p2 * sg.p0 + t) * sg.p1;
        return true;
    }
    const TokPos tp = tok_decode(m, sg.in_tok);
    const int s = sg.mode == LWDETR_OUT_DECONV2x2 ? 2 : 1;      // DECONV: top-left output pixel of this input pixel
    off = (long)tp.b * sg.out_batch_stride + (sg.out_row_offset + tok_encode(0, s * tp.y, s * tp.x, sg.out_tok)) * sg.ldo;
    return tp.valid;
}
__device__ __forceinline__ long col_offset(const lwdetr_gemm_seg& sg, int nl) {
    if (sg.mode == LWDETR_OUT_HEADS) { const int h = nl / sg.p1; return (long)h * sg.p0 * sg.p1 + (nl - h * sg.p1); }
    if (sg.mode == LWDETR_OUT_DECONV2x2) {
        const int q4 = nl / sg.p0, co = nl - q4 * sg.p0;
        return ((long)(q4 >> 1) * sg.out_tok.Wp + (q4 & 1)) * sg.ldo + co;
    }
    return nl;
}

template <typename T, int NV>
__device__ __forceinline__ void store_run(T* dst, const float* x, int cnt, bool aligned) {
    if (cnt >= NV && aligned) {
        typedef T VT __attribute__((ext_vector_type(NV)));
        VT o;
#pragma unroll
        for (int e = 0; e < NV; ++e) o[e] = from_f32<T>(x[e]);
        *(VT*)dst = o;
    } else {
        for (int e = 0; e < NV && e < cnt; ++e) dst[e] = from_f32<T>(x[e]);
    }
}

// ---- the same epilogue with a LayerNorm folded into the GEMM (round 5; lwdetr_gemm_seg.ln_stats / ln_colsum): acc <- (acc - mean_m * colsum_n) *
// rstd_m before the bias. A COPY of epilogue_finish on purpose: with the LayerNorm arithmetic as a run-time (or even as an extra instantiated)
// branch of the shared function, every kernel that inlines it got a different register allocation and schedule - the 256 x 256 large-tile
// kernel, at its 256-register limit, 15-20 % slower on the plain GEMMs of xlarge (same box, profiles/r5g_gemm_big_regression_bisect.txt).
// The plain kernels keep round 4's function below, byte for byte; only the kernels instantiated for folded GEMMs see this one.
// ---- accumulators -> LDS (f32) -> fused LayerNorm / bias / activation / scale / LayerScale / residual / masks ->
// coalesced 16-byte stores in the destination layout of the tile's column segment.
// Second half of the epilogue, shared by every GEMM kernel: one pass of 64 tile rows, already staged in LDS as f32
// (ROW orientation: stage[row * (BN + 4) + col]; COL orientation (HEADS_T): stage[col * 68 + row]), is finished by all
// NTHR threads of the workgroup in runs of 8 consecutive outputs (4 for HEADS_T) - coalesced 16-byte global stores; the
// mode / activation logic lives in a small loop instead of being replicated per accumulator register.
template <typename T, int BN, int NTHR, int FAST_GROUP = 2>   // FAST_GROUP: sweeps the fast path keeps in flight (registers)
__device__ __forceinline__ void epilogue_finish_ln(const lwdetr_gemm_desc& d, const lwdetr_gemm_seg& sg, bool col_orient,
                                                const float* stage, long mbase, int n0) {
    typedef typename Vec<T>::v8 V8;
    const int tid = threadIdx.x;
    constexpr int SLD = BN + 4, SLD_T = 64 + 4;
    const int n_end = sg.n_end < d.N ? sg.n_end : d.N;
    T* __restrict__ out = (T*)sg.out;
    {
        if (!col_orient) {
            // thread -> fixed 8-column run (col), rows strided by 256 / CPRW: column parameters are loop invariant and
            // fetched with two 16-byte loads each (bias / gamma buffers are padded to a multiple of 8 floats by the host)
            constexpr int CPRW = BN / 8, RSTEP = NTHR / CPRW;       // threads per row, rows per sweep (floor: BN = 192 leaves 8 idle)
            T* __restrict__ out2 = (T*)sg.out2;
            const T* __restrict__ res = (const T*)sg.res;
            const int col = (tid % CPRW) * 8, n = n0 + col;
            if (n < n_end && tid < RSTEP * CPRW) {
                const int nl = n - sg.n_begin, cnt = n_end - n < 8 ? n_end - n : 8;
                float bv[8], gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { bv[e] = 0.f; gv[e] = 1.f; }
                if (sg.bias) {
                    const f32x4 b0 = *(const f32x4*)(sg.bias + nl), b1 = *(const f32x4*)(sg.bias + nl + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
                }
                if (sg.gamma) {
                    const f32x4 g0 = *(const f32x4*)(sg.gamma + nl), g1 = *(const f32x4*)(sg.gamma + nl + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gv[e] = g0[e]; gv[4 + e] = g1[e]; }
                }
                const float* __restrict__ lnst = sg.ln_stats;          // planar (2, M): mean, rstd
                float cs[8];
                {
                    const f32x4 c0 = *(const f32x4*)(sg.ln_colsum + nl), c1 = *(const f32x4*)(sg.ln_colsum + nl + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { cs[e] = c0[e]; cs[4 + e] = c1[e]; }
                }
                const long coff = col_offset(sg, nl);
                const int act = sg.act;
                const float scale = sg.scale;
                // Fast path (workgroup-uniform test): all 64 rows exist, whole 16-byte runs, no masks / second destination,
                // separable LINEAR / HEADS addressing. Straight-line code - the stage reads and residual loads of all
                // sweeps are issued together, then the arithmetic, then the stores; the general loop below branches on mode,
                // masks and alignment in every sweep, which serialises the sweeps behind each other's memory latency
                // (measured on the 256 x 256 tile kernel: 9-12 us of epilogue per tile).
                constexpr int ITS = 64 / RSTEP;
                const bool fast = 64 % RSTEP == 0 && mbase + 64 <= d.M && !sg.rowmask && !out2 && (n_end - n0) % 8 == 0 &&
                                  (sg.mode == LWDETR_OUT_LINEAR || sg.mode == LWDETR_OUT_HEADS) && sg.ldo % 8 == 0 &&
                                  ((size_t)out & 15) == 0 && (sg.mode == LWDETR_OUT_LINEAR || sg.p1 % 8 == 0) && sg.n_begin % 8 == 0 &&
                                  (!res || (sg.ldres % 8 == 0 && ((size_t)res & 15) == 0));
                if (fast) {
                    constexpr int G = ITS < FAST_GROUP ? (ITS > 0 ? ITS : 1) : FAST_GROUP;
                    auto finish_all = [&](auto act_tag) {
                        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll 1
                        for (int it0 = 0; it0 < ITS; it0 += G) {
                            f32x4 a0[G], a1[G];
                            V8 rv[G];
                            long ro[G];
                            float lmean[G], lrstd[G];
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                const int row = tid / CPRW + (it0 + g) * RSTEP;
                                const long m = mbase + row;
                                lmean[g] = lnst[m]; lrstd[g] = lnst[(long)d.M + m];
                                if (sg.mode == LWDETR_OUT_LINEAR) ro[g] = m * sg.ldo;
                                else { const int b = (int)(m / sg.p0), t = (int)(m - (long)b * sg.p0); ro[g] = ((long)b * sg.p2 * sg.p0 + t) * sg.p1; }
                                a0[g] = *(const f32x4*)(stage + row * SLD + col);
                                a1[g] = *(const f32x4*)(stage + row * SLD + col + 4);
                            }
                            if (res) {
#pragma unroll
                                for (int g = 0; g < G; ++g) {
                                    const long m = mbase + tid / CPRW + (it0 + g) * RSTEP;
                                    rv[g] = *(const V8*)(res + (sg.res_mod > 0 ? m % sg.res_mod : m) * sg.ldres + nl);
                                }
                            }
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                float x[8];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    x[e] = fmaf(fmaf(-lmean[g], cs[e], a0[g][e]), lrstd[g], bv[e]);
                                    x[4 + e] = fmaf(fmaf(-lmean[g], cs[4 + e], a1[g][e]), lrstd[g], bv[4 + e]);
                                }
                                if (ACT != ACT_NONE) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) x[e] = act_apply<T>(x[e], ACT);
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) x[e] = x[e] * scale * gv[e];
                                if (res) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(rv[g][e]);
                                }
                                V8 o;
#pragma unroll
                                for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(x[e]);
                                *(V8*)(out + ro[g] + coff) = o;
                            }
                        }
                    };
                    // a LayerNorm in front of a Linear: no activation (QKV) or GELU (fc1); anything else takes the general loop
                    if (act == ACT_NONE) { finish_all(std::integral_constant<int, ACT_NONE>{}); return; }
                    if (act == ACT_GELU) { finish_all(std::integral_constant<int, ACT_GELU>{}); return; }
                }
#pragma unroll
                for (int it = 0; it < (64 + RSTEP - 1) / RSTEP; ++it) {
                    const int row = tid / CPRW + it * RSTEP;
                    const long m = mbase + row;
                    long roff;
                    if (row >= 64 || m >= d.M || !row_offset(sg, m, roff)) continue;
                    bool keep_acc = true, keep_out = true;
                    if (sg.rowmask) {
                        const bool rm = sg.rowmask[m] != 0;
                        keep_acc = rm || sg.rowmask_after; keep_out = rm || !sg.rowmask_after;
                    }
                    float x[8];
                    {
                        const f32x4 a0 = *(const f32x4*)(stage + row * SLD + col);
                        const f32x4 a1 = *(const f32x4*)(stage + row * SLD + col + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { x[e] = keep_acc ? a0[e] : 0.f; x[4 + e] = keep_acc ? a1[e] : 0.f; }
                    }
                    {
                        const float mean = lnst[m], rstd = lnst[(long)d.M + m];
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = keep_acc ? fmaf(-mean, cs[e], x[e]) * rstd : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += bv[e];
                    if (act != ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = act_apply<T>(x[e], act);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = x[e] * scale * gv[e];
                    if (res) {
                        const T* rp = res + (sg.res_mod > 0 ? m % sg.res_mod : m) * sg.ldres + nl;
                        if (cnt == 8 && ((size_t)rp & 15) == 0) {
                            const V8 rv = *(const V8*)rp;
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(rv[e]);
                        } else {
                            for (int e = 0; e < cnt; ++e) x[e] += to_f32<T>(rp[e]);
                        }
                    }
                    if (!keep_out) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = 0.f;
                    }
                    T* dst = out + roff + coff;
                    store_run<T, 8>(dst, x, cnt, ((size_t)dst & 15) == 0);
                    if (out2) { T* d2 = out2 + m * sg.ld2 + nl; store_run<T, 8>(d2, x, cnt, ((size_t)d2 & 15) == 0); }
                }
            }
        } else {
            // HEADS_T: out[((b*heads+h)*hd+dd)*Tp + t]: runs of 4 consecutive tokens of one output column. A thread keeps
            // its token run for the whole sweep when NTHR is a multiple of the 16 runs of a column (image index and token
            // offset - two divisions - are computed once; round 1 redid them, in 64 bits, for every run).
            constexpr int RPC = 64 / 4;
            if (NTHR % RPC == 0) {
                const int row = (tid % RPC) * 4;
                const long m = mbase + row;
                if (m < d.M) {
                    const int cnt = d.M - m < 4 ? (int)(d.M - m) : 4;
                    const int b = (int)(m / sg.p0), tk = (int)(m - (long)b * sg.p0);
                    T* obase = out + (long)b * sg.p2 * sg.p1 * sg.p0 + tk;
                    const bool al = (((size_t)obase | ((size_t)sg.p0 * sizeof(T))) & (4 * sizeof(T) - 1)) == 0;
                    const int act = sg.act;
                    const float scale = sg.scale;
                    float lm[4] = {0.f, 0.f, 0.f, 0.f}, lr[4] = {1.f, 1.f, 1.f, 1.f};     // this thread's 4 rows
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (e < cnt) { lm[e] = sg.ln_stats[m + e]; lr[e] = sg.ln_stats[(long)d.M + m + e]; }
#pragma unroll 4
                    for (int coln = tid / RPC; coln < BN; coln += NTHR / RPC) {
                        const int n = n0 + coln;
                        if (n >= n_end) break;
                        const int nl = n - sg.n_begin;
                        const f32x4 a0 = *(const f32x4*)(stage + coln * SLD_T + row);
                        const float bias = sg.bias ? sg.bias[nl] : 0.f;
                        const float csn = sg.ln_colsum[nl];
                        float x[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = act_apply<T>(fmaf(fmaf(-lm[e], csn, a0[e]), lr[e], bias), act) * scale;
                        store_run<T, 4>(obase + (long)nl * sg.p0, x, cnt, al);
                    }
                }
            } else {
#pragma unroll 1
                for (int c = tid; c < BN * RPC; c += NTHR) {
                    const int coln = c / RPC, row = (c - coln * RPC) * 4;
                    const long m = mbase + row;
                    const int n = n0 + coln;
                    if (m >= d.M || n >= n_end) continue;
                    const int nl = n - sg.n_begin, cnt = d.M - m < 4 ? (int)(d.M - m) : 4;
                    const f32x4 a0 = *(const f32x4*)(stage + coln * SLD_T + row);
                    const float bias = sg.bias ? sg.bias[nl] : 0.f;
                    const float csn = sg.ln_colsum[nl];
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float mean = 0.f, rstd = 1.f;
                        if (e < cnt) { mean = sg.ln_stats[m + e]; rstd = sg.ln_stats[(long)d.M + m + e]; }
                        x[e] = act_apply<T>(fmaf(fmaf(-mean, csn, a0[e]), rstd, bias), sg.act) * sg.scale;
                    }
                    const int b = (int)(m / sg.p0), tk = (int)(m - (long)b * sg.p0);
                    T* dst = out + ((long)b * sg.p2 * sg.p1 + nl) * sg.p0 + tk;
                    store_run<T, 4>(dst, x, cnt, ((size_t)dst & (4 * sizeof(T) - 1)) == 0);
                }
            }
        }
    }
}


// ---- shared epilogue: accumulators -> LDS (f32) -> fused bias / activation / scale / LayerScale / residual / masks ->
// coalesced 16-byte stores in the destination layout of the tile's column segment.
// Second half of the epilogue, shared by every GEMM kernel: one pass of 64 tile rows, already staged in LDS as f32
// (ROW orientation: stage[row * (BN + 4) + col]; COL orientation (HEADS_T): stage[col * 68 + row]), is finished by all
// NTHR threads of the workgroup in runs of 8 consecutive outputs (4 for HEADS_T) - coalesced 16-byte global stores; the
// mode / activation logic lives in a small loop instead of being replicated per accumulator register.
template <typename T, int BN, int NTHR, int FAST_GROUP = 2>   // FAST_GROUP: sweeps the fast path keeps in flight (registers)
__device__ __forceinline__ void epilogue_finish(const lwdetr_gemm_desc& d, const lwdetr_gemm_seg& sg, bool col_orient,
                                                const float* stage, long mbase, int n0) {
    typedef typename Vec<T>::v8 V8;
    const int tid = threadIdx.x;
    constexpr int SLD = BN + 4, SLD_T = 64 + 4;
    const int n_end = sg.n_end < d.N ? sg.n_end : d.N;
    T* __restrict__ out = (T*)sg.out;
    {
        if (!col_orient) {
            // thread -> fixed 8-column run (col), rows strided by 256 / CPRW: column parameters are loop invariant and
            // fetched with two 16-byte loads each (bias / gamma buffers are padded to a multiple of 8 floats by the host)
            constexpr int CPRW = BN / 8, RSTEP = NTHR / CPRW;       // threads per row, rows per sweep (floor: BN = 192 leaves 8 idle)
            T* __restrict__ out2 = (T*)sg.out2;
            const T* __restrict__ res = (const T*)sg.res;
            const int col = (tid % CPRW) * 8, n = n0 + col;
            if (n < n_end && tid < RSTEP * CPRW) {
                const int nl = n - sg.n_begin, cnt = n_end - n < 8 ? n_end - n : 8;
                float bv[8], gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { bv[e] = 0.f; gv[e] = 1.f; }
                if (sg.bias) {
                    const f32x4 b0 = *(const f32x4*)(sg.bias + nl), b1 = *(const f32x4*)(sg.bias + nl + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
                }
                if (sg.gamma) {
                    const f32x4 g0 = *(const f32x4*)(sg.gamma + nl), g1 = *(const f32x4*)(sg.gamma + nl + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gv[e] = g0[e]; gv[4 + e] = g1[e]; }
                }
                const long coff = col_offset(sg, nl);
                const int act = sg.act;
                const float scale = sg.scale;
                // Fast path (workgroup-uniform test): all 64 rows exist, whole 16-byte runs, no masks / second destination,
                // separable LINEAR / HEADS addressing. Straight-line code - the stage reads and residual loads of all
                // sweeps are issued together, then the arithmetic, then the stores; the general loop below branches on mode,
                // masks and alignment in every sweep, which serialises the sweeps behind each other's memory latency
                // (measured on the 256 x 256 tile kernel: 9-12 us of epilogue per tile).
                constexpr int ITS = 64 / RSTEP;
                const bool fast = 64 % RSTEP == 0 && mbase + 64 <= d.M && !sg.rowmask && !out2 && (n_end - n0) % 8 == 0 &&
                                  (sg.mode == LWDETR_OUT_LINEAR || sg.mode == LWDETR_OUT_HEADS) && sg.ldo % 8 == 0 &&
                                  ((size_t)out & 15) == 0 && (sg.mode == LWDETR_OUT_LINEAR || sg.p1 % 8 == 0) && sg.n_begin % 8 == 0 &&
                                  (!res || (sg.ldres % 8 == 0 && ((size_t)res & 15) == 0));
                if (fast) {
                    constexpr int G = ITS < FAST_GROUP ? (ITS > 0 ? ITS : 1) : FAST_GROUP;
                    auto finish_all = [&](auto act_tag) {
                        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll 1
                        for (int it0 = 0; it0 < ITS; it0 += G) {
                            f32x4 a0[G], a1[G];
                            V8 rv[G];
                            long ro[G];
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                const int row = tid / CPRW + (it0 + g) * RSTEP;
                                const long m = mbase + row;
                                if (sg.mode == LWDETR_OUT_LINEAR) ro[g] = m * sg.ldo;
                                else { const int b = (int)(m / sg.p0), t = (int)(m - (long)b * sg.p0); ro[g] = ((long)b * sg.p2 * sg.p0 + t) * sg.p1; }
                                a0[g] = *(const f32x4*)(stage + row * SLD + col);
                                a1[g] = *(const f32x4*)(stage + row * SLD + col + 4);
                            }
                            if (res) {
#pragma unroll
                                for (int g = 0; g < G; ++g) {
                                    const long m = mbase + tid / CPRW + (it0 + g) * RSTEP;
                                    rv[g] = *(const V8*)(res + (sg.res_mod > 0 ? m % sg.res_mod : m) * sg.ldres + nl);
                                }
                            }
#pragma unroll
                            for (int g = 0; g < G; ++g) {
                                float x[8];
#pragma unroll
                                for (int e = 0; e < 4; ++e) { x[e] = a0[g][e] + bv[e]; x[4 + e] = a1[g][e] + bv[4 + e]; }
                                if (ACT != ACT_NONE) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) x[e] = act_apply<T>(x[e], ACT);
                                }
#pragma unroll
                                for (int e = 0; e < 8; ++e) x[e] = x[e] * scale * gv[e];
                                if (res) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(rv[g][e]);
                                }
                                V8 o;
#pragma unroll
                                for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(x[e]);
                                *(V8*)(out + ro[g] + coff) = o;
                            }
                        }
                    };
                    if (act == ACT_NONE) finish_all(std::integral_constant<int, ACT_NONE>{});
                    else if (act == ACT_GELU) finish_all(std::integral_constant<int, ACT_GELU>{});
                    else if (act == ACT_SILU) finish_all(std::integral_constant<int, ACT_SILU>{});
                    else finish_all(std::integral_constant<int, ACT_RELU>{});
                    return;
                }
#pragma unroll
                for (int it = 0; it < (64 + RSTEP - 1) / RSTEP; ++it) {
                    const int row = tid / CPRW + it * RSTEP;
                    const long m = mbase + row;
                    long roff;
                    if (row >= 64 || m >= d.M || !row_offset(sg, m, roff)) continue;
                    bool keep_acc = true, keep_out = true;
                    if (sg.rowmask) {
                        const bool rm = sg.rowmask[m] != 0;
                        keep_acc = rm || sg.rowmask_after; keep_out = rm || !sg.rowmask_after;
                    }
                    float x[8];
                    {
                        const f32x4 a0 = *(const f32x4*)(stage + row * SLD + col);
                        const f32x4 a1 = *(const f32x4*)(stage + row * SLD + col + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { x[e] = keep_acc ? a0[e] : 0.f; x[4 + e] = keep_acc ? a1[e] : 0.f; }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += bv[e];
                    if (act != ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = act_apply<T>(x[e], act);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = x[e] * scale * gv[e];
                    if (res) {
                        const T* rp = res + (sg.res_mod > 0 ? m % sg.res_mod : m) * sg.ldres + nl;
                        if (cnt == 8 && ((size_t)rp & 15) == 0) {
                            const V8 rv = *(const V8*)rp;
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(rv[e]);
                        } else {
                            for (int e = 0; e < cnt; ++e) x[e] += to_f32<T>(rp[e]);
                        }
                    }
                    if (!keep_out) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] = 0.f;
                    }
                    T* dst = out + roff + coff;
                    store_run<T, 8>(dst, x, cnt, ((size_t)dst & 15) == 0);
                    if (out2) { T* d2 = out2 + m * sg.ld2 + nl; store_run<T, 8>(d2, x, cnt, ((size_t)d2 & 15) == 0); }
                }
            }
        } else {
            // HEADS_T: out[((b*heads+h)*hd+dd)*Tp + t]: runs of 4 consecutive tokens of one output column. A thread keeps
            // its token run for the whole sweep when NTHR is a multiple of the 16 runs of a column (image index and token
            // offset - two divisions - are computed once; round 1 redid them, in 64 bits, for every run).
            constexpr int RPC = 64 / 4;
            if (NTHR % RPC == 0) {
                const int row = (tid % RPC) * 4;
                const long m = mbase + row;
                if (m < d.M) {
                    const int cnt = d.M - m < 4 ? (int)(d.M - m) : 4;
                    const int b = (int)(m / sg.p0), tk = (int)(m - (long)b * sg.p0);
                    T* obase = out + (long)b * sg.p2 * sg.p1 * sg.p0 + tk;
                    const bool al = (((size_t)obase | ((size_t)sg.p0 * sizeof(T))) & (4 * sizeof(T) - 1)) == 0;
                    const int act = sg.act;
                    const float scale = sg.scale;
#pragma unroll 4
                    for (int coln = tid / RPC; coln < BN; coln += NTHR / RPC) {
                        const int n = n0 + coln;
                        if (n >= n_end) break;
                        const int nl = n - sg.n_begin;
                        const f32x4 a0 = *(const f32x4*)(stage + coln * SLD_T + row);
                        const float bias = sg.bias ? sg.bias[nl] : 0.f;
                        float x[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[e] = act_apply<T>(a0[e] + bias, act) * scale;
                        store_run<T, 4>(obase + (long)nl * sg.p0, x, cnt, al);
                    }
                }
            } else {
#pragma unroll 1
                for (int c = tid; c < BN * RPC; c += NTHR) {
                    const int coln = c / RPC, row = (c - coln * RPC) * 4;
                    const long m = mbase + row;
                    const int n = n0 + coln;
                    if (m >= d.M || n >= n_end) continue;
                    const int nl = n - sg.n_begin, cnt = d.M - m < 4 ? (int)(d.M - m) : 4;
                    const f32x4 a0 = *(const f32x4*)(stage + coln * SLD_T + row);
                    const float bias = sg.bias ? sg.bias[nl] : 0.f;
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = act_apply<T>(a0[e] + bias, sg.act) * sg.scale;
                    const int b = (int)(m / sg.p0), tk = (int)(m - (long)b * sg.p0);
                    T* dst = out + ((long)b * sg.p2 * sg.p1 + nl) * sg.p0 + tk;
                    store_run<T, 4>(dst, x, cnt, ((size_t)dst & (4 * sizeof(T) - 1)) == 0);
                }
            }
        }
    }
}

// ---- epilogue of the 4-wave (2 x 2) kernels: accumulators -> LDS (f32, 64 tile rows per pass) -> epilogue_finish
template <typename T, int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(const lwdetr_gemm_desc& d, const lwdetr_gemm_seg& sg, bool col_orient,
                                              f32x4 (&acc)[BN / 32][BM / 32], T* smem, long m0, int n0) {
    constexpr int WM = BM / 2, WN = BN / 2, TT = WM / 16, FT = WN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    float* stage = (float*)smem;
    constexpr int SLD = BN + 4, SLD_T = 64 + 4;
    for (int pass = 0; pass < BM / 64; ++pass) {
        if (BM == 64 || wm == pass) {
            const int r0 = BM == 64 ? wm * WM : 0;          // first stage row of this wave
            float* sp = col_orient ? stage + (wn * WN + l15) * SLD_T + r0 + g * 4 : stage + (r0 + l15) * SLD + wn * WN + g * 4;
            const int fs = col_orient ? 16 * SLD_T : 16, ts = col_orient ? 16 : 16 * SLD;
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int t = 0; t < TT; ++t) *(f32x4*)(sp + f * fs + t * ts) = acc[f][t];
        }
        __syncthreads();
        epilogue_finish<T, BN, 256>(d, sg, col_orient, stage, m0 + pass * 64, n0);
        __syncthreads();
    }
}

template <typename T, int BM, int BN, int AMODE>
__global__ __launch_bounds__(256) void gemm_kernel(const lwdetr_gemm_desc d) {
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int CPR = BK / EPC;              // chunks per tile row
    constexpr int LDS_LD = BK + 2 * EPC;       // +32 B: row stride 6 (16-bit) / 10 (f32) sixteen-byte slots, == 2 (mod 4):
                                               // conflict-free for the 16-lane ds_read_b128 service groups
    constexpr int RPP = 256 / CPR;             // tile rows covered by one pass of the 256 threads
    constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
    constexpr int WM = BM / 2, WN = BN / 2, TT = WM / 16, FT = WN / 16;
    typedef typename Vec<T>::v8 V8;

    __shared__ __attribute__((aligned(16))) T smem[2 * (BM + BN) * LDS_LD];

    // ---- XCD-aware, bijective workgroup remap (block b runs on XCD b % 8): consecutive logical tiles share an XCD
    const int tiles_n = (d.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int lc = tid % CPR, lr = tid / CPR;

    const T* __restrict__ A = (const T*)d.A;
    const T* __restrict__ A2 = (const T*)d.A2;
    const T* __restrict__ W = (const T*)d.W;

    // ---- per-thread A row descriptors (fixed for the whole k loop)
    const T* a_row[A_PASSES];
    int a_b[A_PASSES], a_y[A_PASSES], a_x[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const long m = m0 + i * RPP + lr;
        a_row[i] = nullptr; a_b[i] = -1; a_y[i] = 0; a_x[i] = 0;
        if (m < d.M) {
            if (AMODE == LWDETR_A_PLAIN) {
                a_row[i] = A + m * d.lda;
            } else if (AMODE == LWDETR_A_CONV3x3) {
                const int hw = d.conv_hout * d.conv_wout;
                const int b = (int)(m / hw), r = (int)(m - (long)b * hw);
                a_b[i] = b; a_y[i] = r / d.conv_wout; a_x[i] = r - a_y[i] * d.conv_wout;
            } else {
                const TokPos p = tok_decode(m, d.a_tok);
                if (p.valid) { a_b[i] = p.b; a_y[i] = p.y; a_x[i] = p.x; }
            }
        }
    }
    const T* w_row[B_PASSES];
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
        const int n = n0 + i * RPP + lr;
        w_row[i] = n < d.N ? W + (long)n * d.K : nullptr;
    }

    uint4 ra[A_PASSES], rb[B_PASSES];

    // Loads are unconditional from an always-valid address and masked afterwards: a `cond ? *p : 0` form makes the
    // compiler select between p and a zero living in scratch (flat loads + private memory).
    auto masked = [&](const uint4 v, bool ok) {
        return make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
    };
    auto load_tile = [&](int kt) {
        const int k = kt * BK + lc * EPC;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const T* src = A;
            bool ok = false;
            if (AMODE == LWDETR_A_PLAIN) {
                ok = a_row[i] != nullptr;
                if (ok) src = a_row[i] + k;
            } else if (AMODE == LWDETR_A_CONV3x3) {
                const int tap = (kt * BK) / d.conv_cin;             // uniform: Cin % 32 == 0
                const int ci = k - tap * d.conv_cin;
                const int iy = a_y[i] * d.conv_stride + tap / 3 - 1, ix = a_x[i] * d.conv_stride + tap % 3 - 1;
                ok = a_b[i] >= 0 && iy >= 0 && iy < d.a_tok.Hp && ix >= 0 && ix < d.a_tok.Wp;
                if (ok) src = A + tok_encode(a_b[i], iy, ix, d.a_tok) * d.lda + d.a_col0 + ci;
            } else {
                ok = a_b[i] >= 0;
                const int ch = k >> 8, py = (k >> 4) & 15, px = k & 15;
                if (ok) src = A + (((long)a_b[i] * 3 + ch) * d.img_h + a_y[i] * 16 + py) * d.img_w + a_x[i] * 16 + px;
            }
            uint4 v = *(const uint4*)src;
            if (AMODE == LWDETR_A_PLAIN && A2) {
                typedef T VC __attribute__((ext_vector_type(EPC)));
                VC va = __builtin_bit_cast(VC, v);
                const VC vb = __builtin_bit_cast(VC, *(const uint4*)(A2 + (src - A)));
#pragma unroll
                for (int e = 0; e < EPC; ++e) va[e] = from_f32<T>(to_f32<T>(va[e]) + to_f32<T>(vb[e]));
                v = __builtin_bit_cast(uint4, va);
            }
            ra[i] = masked(v, ok);
        }
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const bool ok = w_row[i] != nullptr;
            rb[i] = masked(*(const uint4*)((ok ? w_row[i] : W) + k), ok);
        }
    };
    auto store_tile = [&](int buf) {
        T* As = smem + buf * (BM + BN) * LDS_LD;
        T* Bs = As + BM * LDS_LD;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) *(uint4*)(As + (i * RPP + lr) * LDS_LD + lc * EPC) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) *(uint4*)(Bs + (i * RPP + lr) * LDS_LD + lc * EPC) = rb[i];
    };

    // ---- segment of this column tile (segment boundaries are multiples of BN, checked by the launcher)
    int si = 0;
#pragma unroll
    for (int s = 1; s < 3; ++s) if (s < d.nseg && n0 >= d.seg[s].n_begin) si = s;
    const lwdetr_gemm_seg& sg = d.seg[si];
    const bool col_orient = sg.mode == LWDETR_OUT_HEADS_T;

    f32x4 acc[FT][TT];
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = d.K / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const T* As = smem + buf * (BM + BN) * LDS_LD;
        const T* Bs = As + BM * LDS_LD;
        V8 xf[TT], wf[FT];
#pragma unroll
        for (int t = 0; t < TT; ++t) xf[t] = *(const V8*)(As + (wm * WM + t * 16 + l15) * LDS_LD + g * 8);
#pragma unroll
        for (int f = 0; f < FT; ++f) wf[f] = *(const V8*)(Bs + (wn * WN + f * 16 + l15) * LDS_LD + g * 8);
        if (!col_orient) {
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int t = 0; t < TT; ++t) acc[f][t] = Mma<T>::k32(wf[f], xf[t], acc[f][t]);
        } else {
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int t = 0; t < TT; ++t) acc[f][t] = Mma<T>::k32(xf[t], wf[f], acc[f][t]);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    gemm_epilogue<T, BM, BN>(d, sg, col_orient, acc, smem, m0, n0);
}

// ---- DMA-staged, multi-stage variant (16-bit types, no A2): operand tiles go global -> LDS with global_load_lds (no
// staging registers, no ds_write pass) through an NST-deep ring of BK = 32 stages, NST-1 tiles in flight: counted
// s_waitcnt vmcnt + raw s_barrier keep the prefetches alive across barriers (a __syncthreads would drain them). Every
// wave issues the same number of DMA pieces per tile (dummy pieces read a zero page past the end of K), so the wait
// count is a compile-time constant. The LDS image is unpadded with an XOR swizzle (16-byte slot c of row r lives at
// slot c ^ ((r >> 1) & 3)) - conflict-free for the 16-lane ds_read_b128 service groups; a DMA piece is lane-linear in
// LDS, so the swizzle is applied to the SOURCE address.
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];      // zero page: source of masked (out-of-range) lanes

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one 1 KB DMA piece, untracked by hipcc (inline assembly; drained by counted waits): lane l's 16 bytes land at the wave-uniform
// LDS address lds_wave_base + 16 l
__device__ __forceinline__ void gdma16(const void* src, const void* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(src) : "memory");
}

// KB = k-depth of a stage: 32 (a DMA piece = 16 rows x 64 B) or 64 (8 rows x 128 B: whole cache lines, half the
// barriers / waits / address updates per byte; slot c of row r lives at slot c ^ ((r >> 1) & (KB / 8 - 1))).
// SPLITK (round 5; 64 x 64 tiles only): few-row GEMMs with a long contraction (one image: the 3x3 convolutions of the projector, M = 1600,
// K = 1152 on 50 tiles - 36 dependent DMA -> barrier -> fragment -> MFMA steps, 19 us where the matrix work is < 1 us) run d.splitk
// workgroups per tile, each over a contiguous range of k-stages. A workgroup stages its partial tile in LDS as usual, writes it to its
// slab of d.splitk_ws (f32), publishes it (agent-scope release fence, then a relaxed ticket on the tile's counter) and leaves; the one that
// draws the last ticket acquires, sums ALL slabs of the tile in slice order (its own included, from memory: the f32 sum does not depend on
// who arrives last - results stay bit-reproducible), resets the counter and runs the epilogue. Slices of a tile are neighbours in the
// (XCD-remapped) workgroup order, so the slabs are read out of the reducer's own L2.
template <typename T, int BM, int BN, int AMODE, int NST, int KB = 32, bool SPLITK = false>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const lwdetr_gemm_desc d) {
    static_assert(!SPLITK || (BM == 64 && BN == 64), "split-K: 64 x 64 tiles");
    constexpr int EPC = 8;
    constexpr int SLOTS = KB / EPC, RP = 64 / SLOTS, KC = KB / 32;   // 16-byte slots per row, rows per DMA piece, MFMA k-chunks
    constexpr int A_MY = BM / RP / 4, B_MY = BN / RP / 4;        // DMA pieces (64 slots) per wave and operand
    constexpr int PER_TILE = A_MY + B_MY;
    constexpr int WM = BM / 2, WN = BN / 2, TT = WM / 16, FT = WN / 16;
    constexpr int STAGE = (BM + BN) * KB;
    typedef typename Vec<T>::v8 V8;
    static_assert(sizeof(T) == 2, "DMA variant is instantiated for f16 / bf16");
    constexpr int LDS_ELEMS = NST * STAGE > 64 * (BN + 4) * 2 ? NST * STAGE : 64 * (BN + 4) * 2;   // ring, or f32 staging
    __shared__ __attribute__((aligned(16))) T smem[LDS_ELEMS];

    const int tiles_n = (d.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nsl = SPLITK ? d.splitk : 1;
    const int tile = SPLITK ? wg / nsl : wg, slice = SPLITK ? wg - tile * nsl : 0;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const T* __restrict__ A = (const T*)d.A;
    const T* __restrict__ W = (const T*)d.W;
    const T* zero = (const T*)g_zero16;

    // lane -> (row inside a piece, physical slot); the logical 16-byte column a lane fetches is the slot un-swizzled with
    // the row's key (piece i starts at row RP * i, so for KB = 64 the key also depends on the parity of i)
    const int prow = lane / SLOTS, pslot = lane % SLOTS;
    auto ccol_of = [&](int piece) { return (pslot ^ ((((piece * RP) + prow) >> 1) & (SLOTS - 1))) * EPC; };
    const T* a_src[A_MY]; int a_b[A_MY], a_y[A_MY], a_x[A_MY], a_cc[A_MY];
#pragma unroll
    for (int k = 0; k < A_MY; ++k) {
        const long m = m0 + RP * (wave + 4 * k) + prow;
        const int ccol = ccol_of(wave + 4 * k);
        a_src[k] = nullptr; a_b[k] = -1; a_y[k] = 0; a_x[k] = 0; a_cc[k] = ccol;
        if (m < d.M) {
            if (AMODE == LWDETR_A_PLAIN) a_src[k] = A + m * d.lda + ccol;
            else if (AMODE == LWDETR_A_CONV3x3) {
                const int hw = d.conv_hout * d.conv_wout;
                const int b = (int)(m / hw), r = (int)(m - (long)b * hw);
                a_b[k] = b; a_y[k] = r / d.conv_wout; a_x[k] = r - a_y[k] * d.conv_wout;
            } else {
                const TokPos tp = tok_decode(m, d.a_tok);
                if (tp.valid) { a_b[k] = tp.b; a_y[k] = tp.y; a_x[k] = tp.x; }
            }
        }
    }
    const T* w_src[B_MY];
#pragma unroll
    for (int k = 0; k < B_MY; ++k) {
        const int n = n0 + RP * (wave + 4 * k) + prow;
        w_src[k] = n < d.N ? W + (long)n * d.K + ccol_of(wave + 4 * k) : nullptr;
    }
    const int nk_all = d.K / KB;
    const int kt_lo = SPLITK ? (int)((long)slice * nk_all / nsl) : 0, kt_hi = SPLITK ? (int)((long)(slice + 1) * nk_all / nsl) : nk_all;
    const int nk = kt_hi - kt_lo;        // this workgroup's k-stages: [kt_lo, kt_hi)
    const T* conv_src[A_MY]; int conv_tap[A_MY];
#pragma unroll
    for (int k = 0; k < A_MY; ++k) { conv_src[k] = nullptr; conv_tap[k] = -1; }
    auto stage = [&](int kt) {           // always PER_TILE pieces; tiles past the end of K read the zero page
        T* As = smem + (kt % NST) * STAGE;
        T* Bs = As + BM * KB;
        const int k0 = (kt_lo + kt) * KB;
        const bool live = kt < nk;
#pragma unroll
        for (int k = 0; k < A_MY; ++k) {
            const T* src = zero;
            if (AMODE == LWDETR_A_PLAIN) {
                if (live && a_src[k]) src = a_src[k] + k0;
            } else if (AMODE == LWDETR_A_CONV3x3) {
                // the shifted source row only changes with the tap (every Cin / 32 stages): its address (bounds test,
                // token encode: ~40 VALU instructions per piece) is kept across the stages of a tap
                const int tap = k0 / d.conv_cin;                    // uniform: Cin % KB == 0
                if (tap != conv_tap[k]) {
                    conv_tap[k] = tap;
                    const int iy = a_y[k] * d.conv_stride + tap / 3 - 1, ix = a_x[k] * d.conv_stride + tap % 3 - 1;
                    conv_src[k] = (a_b[k] >= 0 && iy >= 0 && iy < d.a_tok.Hp && ix >= 0 && ix < d.a_tok.Wp)
                                      ? A + tok_encode(a_b[k], iy, ix, d.a_tok) * d.lda + d.a_col0 + a_cc[k] : nullptr;
                }
                if (live && conv_src[k]) src = conv_src[k] + (k0 - tap * d.conv_cin);
            } else {
                const int kk = k0 + a_cc[k], ch = kk >> 8, py = (kk >> 4) & 15, px = kk & 15;
                if (live && a_b[k] >= 0)
                    src = A + (((long)a_b[k] * 3 + ch) * d.img_h + a_y[k] * 16 + py) * d.img_w + a_x[k] * 16 + px;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(As + (wave + 4 * k) * 64 * EPC), 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < B_MY; ++k) {
            const T* src = (live && w_src[k]) ? w_src[k] + k0 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Bs + (wave + 4 * k) * 64 * EPC), 16, 0, 0);
        }
    };

    int si = 0;
#pragma unroll
    for (int s = 1; s < 3; ++s) if (s < d.nseg && n0 >= d.seg[s].n_begin) si = s;
    const lwdetr_gemm_seg& sg = d.seg[si];
    const bool col_orient = sg.mode == LWDETR_OUT_HEADS_T;

    f32x4 acc[FT][TT];
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int pc[KC];                                        // swizzled slots of this lane's k-runs
#pragma unroll
    for (int c = 0; c < KC; ++c) pc[c] = ((c * 4 + g) ^ ((l15 >> 1) & (SLOTS - 1))) * EPC;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) stage(s);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vmcnt<(NST - 2) * PER_TILE>();            // tile kt has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                  // ... and everybody's; everybody also left tile kt-1's buffer
        stage(kt + NST - 1);                           // refill the buffer tile kt-1 used
        const T* As = smem + (kt % NST) * STAGE;
        const T* Bs = As + BM * KB;
        V8 xf[KC][TT], wf[KC][FT];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
#pragma unroll
            for (int t = 0; t < TT; ++t) xf[c][t] = *(const V8*)(As + (wm * WM + t * 16 + l15) * KB + pc[c]);
#pragma unroll
            for (int f = 0; f < FT; ++f) wf[c][f] = *(const V8*)(Bs + (wn * WN + f * 16 + l15) * KB + pc[c]);
        }
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            if (!col_orient) {
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[f][t] = Mma<T>::k32(wf[c][f], xf[c][t], acc[f][t]);
            } else {
#pragma unroll
                for (int f = 0; f < FT; ++f)
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[f][t] = Mma<T>::k32(xf[c][t], wf[c][f], acc[f][t]);
            }
        }
    }
    __syncthreads();            // drains the dummy tail pieces and the last fragment reads before LDS is re-used
    if constexpr (SPLITK) {
        // ---- partial tile -> LDS stage (gemm_epilogue's layout for BM = 64) -> slab -> ticket; the last arriver reduces and finishes
        constexpr int SLD = BN + 4, SLD_T = 64 + 4, SLAB = 64 * 68;         // floats per slab (either orientation)
        float* stg = (float*)smem;
        int* flag = (int*)smem + SLAB + 16;                                 // one word of the same LDS array (no second __shared__ object)
        static_assert((SLAB + 32) * 4 <= LDS_ELEMS * (int)sizeof(T), "stage + flag fit the ring area");
        {
            float* sp = col_orient ? stg + (wn * WN + l15) * SLD_T + wm * WM + g * 4 : stg + (wm * WM + l15) * SLD + wn * WN + g * 4;
            const int fs = col_orient ? 16 * SLD_T : 16, ts = col_orient ? 16 : 16 * SLD;
#pragma unroll
            for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int t = 0; t < TT; ++t) *(f32x4*)(sp + f * fs + t * ts) = acc[f][t];
        }
        __syncthreads();
        float* ws = (float*)d.splitk_ws;
        const long ntiles = (long)gridDim.x / nsl;
        int* cnt = (int*)(ws + ntiles * nsl * SLAB);
        float* slab = ws + ((long)tile * nsl + slice) * SLAB;
        for (int i = tid; i < SLAB / 4; i += 256) ((f32x4*)slab)[i] = ((const f32x4*)stg)[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the write-back is complete before the ticket is drawn (guide G16)
            *flag = __hip_atomic_fetch_add(cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*flag != nsl - 1) return;                                       // workgroup-uniform: not the last slice of this tile
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        const float* s0 = ws + (long)tile * nsl * SLAB;
        for (int i = tid; i < SLAB / 4; i += 256) {
            f32x4 a = ((const f32x4*)s0)[i];
            for (int s2 = 1; s2 < nsl; ++s2) { const f32x4 b = ((const f32x4*)(s0 + (long)s2 * SLAB))[i]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
            ((f32x4*)stg)[i] = a;
        }
        if (tid == 0) __hip_atomic_store(cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next launch starts from zero
        __syncthreads();
        epilogue_finish<T, BN, 256>(d, sg, col_orient, stg, m0, n0);
    } else {
        gemm_epilogue<T, BM, BN>(d, sg, col_orient, acc, smem, m0, n0);
    }
}

template <typename T> struct Mma32;
template <> struct Mma32<f16> {
    static __device__ __forceinline__ f32x16 k16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<bf16> {
    static __device__ __forceinline__ f32x16 k16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

// ---- 3x3 convolution, stride 1, raster (b, y, x) token rows, N = Cin <= 192: the input patch stays in LDS --------------------
// The implicit-GEMM view above re-fetches every input row nine times (once per tap) and a 128 x 128 tile of it pulls
// (128 + 128) x 9 Cin x 2 bytes into its CU for 128 x 128 x 9 Cin MACs: at Cin = 128 that is 590 KB per tile in 64-byte segments
// (27 B / clk / CU, profiles/r2g_ubench_ingress.txt: 9 us of ingress for 3.9 us of MFMA work - the kernel ran at 0.13 of peak).
// Here a workgroup owns 128 consecutive output pixels and ALL output channels. The pixels' input rows plus W + 1 rows of halo on
// either side (the taps of pixel m are rows m + dy W + dx of the same image) are DMA'd into LDS once, as whole rows; the
// fragments of all nine taps are read from that patch at a tap-dependent row offset, and only the weights stream through an
// NST-deep ring of 32-deep stages: 54 KB + 295 KB of ingress per workgroup instead of 590 KB, in contiguous pieces.
// Taps that fall outside the image (first / last row or column; a tile may also straddle two images) are masked in the fragment:
// per lane a 9-bit validity word per 32-row tile, and a wave-uniform "some lane needs it" word so that interior tiles pay nothing.
// 32x32x16 MFMAs (D[n][m] = W fragment x pixel fragment, as in gemm_big_kernel): the 16 lanes a ds_read_b128 serves together then
// hold the SAME k-run of 16 different rows, whatever the tap's row offset, and patch rows padded by one 16-byte slot (row stride
// 17 / 25 slots: odd) put those rows on 16 different banks; a chunk's offset inside the row stays an immediate. (With the
// 16x16x32 layout a service group mixes two k-runs and no row-major image is conflict-free for odd AND even row offsets.)
// Phase timing (tools/conv_ablate.sh timing: -DLWDETR_CONV_TIMING=<workgroup>, never in the product): per wave of that workgroup,
// 10 ns ticks at: patch + first stages issued, masks computed, everything landed, loop done (with the time spent in the lgkmcnt
// wait, the vmcnt wait and the barrier inside it), epilogue done.
#ifdef LWDETR_CONV_TIMING
__device__ unsigned long long g_conv_timing[4][12];
extern "C" int lwdetr_debug_conv_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_timing), sizeof(g_conv_timing)) == hipSuccess ? 0 : -1;
}
#define CONV_NOW() __builtin_amdgcn_s_memrealtime()
#endif
// Tuning builds (tools/conv_ablate.sh, -DLWDETR_CONV_ABL=bits, wrong results): 1 = no MFMAs, 2 = no patch DMA, 4 = no epilogue,
// 8 = no weight DMA, 16 = no fragment reads, 32 = no lgkmcnt wait in front of the stage barrier.
#ifndef LWDETR_CONV_ABL
#define LWDETR_CONV_ABL 0
#endif
// One 16-byte fragment read the compiler does not track (counted lgkmcnt waits below; a tracked read that crosses the back edge
// of the k-loop makes hipcc wait with lgkmcnt(0) in front of the first MFMA of every chunk, i.e. for the reads it has just issued
// for the NEXT chunk: no overlap at all - 1150 cycles per stage for 256 cycles of MFMA, profiles/r3e_conv3x3_patch.txt).
template <typename V8>
__device__ __forceinline__ void lds_read16(V8& dst, unsigned lds_addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr)); }
// n / dvs for n < 2^31 with magic = floor(2^32 / dvs) (the estimate is at most one too small)
__device__ __forceinline__ void divmod_magic(unsigned n, unsigned dvs, unsigned magic, unsigned& q, unsigned& r) {
    q = __umulhi(n, magic); r = n - q * dvs;
    if (r >= dvs) { ++q; r -= dvs; }
}

template <typename T, int CIN, int BN, int NST>
__global__ __launch_bounds__(256) void conv3x3_patch_kernel(const lwdetr_gemm_desc d, int np_patch, unsigned magic_hw, unsigned magic_w) {
    constexpr int BM = 128, KB = 32, EPC = 8;
    constexpr int SPR = CIN / EPC + 1, RS = SPR * EPC;      // slots / elements per patch row (one pad slot)
    constexpr int NCH = CIN / KB;                           // stages per tap
    constexpr int B_MY = BN / 16 / 4;                       // weight pieces (16 rows x 64 B) per wave and stage
    constexpr int WN = BN / 2, TM = 2, TN = WN / 32, NRD = TM + TN;       // fragment reads per 16-deep chunk
    typedef typename Vec<T>::v8 V8;
    static_assert(sizeof(T) == 2 && BN % 64 == 0 && CIN % KB == 0 && (NCH == 4 || NCH == 6), "16-bit, whole pieces, 4 or 6 stages per tap");
    extern __shared__ __attribute__((aligned(16))) unsigned char conv_smem[];
    T* patch = (T*)conv_smem;
    T* ring = patch + (long)np_patch * 512;                 // np_patch pieces of 1 KB
    const unsigned lds_patch = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)conv_smem;
    const unsigned lds_ring = lds_patch + (unsigned)np_patch * 1024u;

    const int nwg = gridDim.x;
    int wg;
    {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;      // neighbouring tiles (shared halo) on one XCD
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = wg * BM;                                 // the launcher bounds M * lda * 2 (and M) below 2^31
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int Wimg = d.conv_wout, Himg = d.conv_hout, PR = BM + 2 * Wimg + 2;
    const T* __restrict__ A = (const T*)d.A + d.a_col0;
    const T* __restrict__ W = (const T*)d.W;
    const T* zero = (const T*)g_zero16;
#ifdef LWDETR_CONV_TIMING
    const unsigned long long ct0 = CONV_NOW(); unsigned long long ct_lgkm = 0, ct_vm = 0, ct_bar = 0;
#endif

    // ---- weight ring: stage image of gemm_big_kernel at KB = 32 (16-byte slot c of row r at slot c ^ ((r >> 2) & 3))
    const int prow = lane >> 2, pslot = lane & 3;
    const T* w_src[B_MY];
#pragma unroll
    for (int k = 0; k < B_MY; ++k) {
        const int n = 16 * (wave + 4 * k) + prow;
        w_src[k] = n < d.N ? W + (long)n * d.K + (pslot ^ ((prow >> 2) & 3)) * EPC : nullptr;
    }
    const int nk = 9 * NCH;
    auto stage = [&](int kt, int slot) {
        T* Bs = ring + slot * (BN * KB);
#pragma unroll
        for (int k = 0; k < B_MY; ++k) {
            if (LWDETR_CONV_ABL & 8) continue;
            gdma16((kt < nk && w_src[k]) ? w_src[k] + kt * KB : zero, Bs + (wave + 4 * k) * 64 * EPC);
        }
    };
#pragma unroll
    for (int s = 0; s < NST; ++s) stage(s, s);
    // ---- the patch: piece p = 64 consecutive 16-byte slots of the padded row image (slot q: row q / SPR, column q % SPR)
    {
        constexpr unsigned MAGIC_SPR = 0x100000000ull / SPR;
        constexpr int DR = 256 / SPR, DS = 256 - DR * SPR;           // a wave's next piece is 256 slots on: DR rows and DS slots
        const int lda = (int)d.lda, gm0 = m0 - Wimg - 1;
        unsigned pr, ps;
        divmod_magic((unsigned)(wave * 64 + lane), SPR, MAGIC_SPR, pr, ps);
        int goff = (gm0 + (int)pr) * lda + (int)ps * EPC;            // element offset of this lane's slot in A
        for (int p = wave; p < np_patch; p += 4) {
            const int gm = gm0 + (int)pr;
            const bool ok = ps < SPR - 1 && (int)pr < PR && gm >= 0 && gm < d.M;
            if (!(LWDETR_CONV_ABL & 2)) gdma16(ok ? A + goff : zero, patch + p * 512);
            pr += DR; ps += DS; goff += DR * lda + DS * EPC;
            if (ps >= SPR) { ps -= SPR; ++pr; goff += lda - SPR * EPC; }
        }
    }
#ifdef LWDETR_CONV_TIMING
    const unsigned long long ct1 = CONV_NOW();
#endif
    // ---- validity of the nine taps for this lane's two pixels; wave-uniform "some lane needs the mask" words
    unsigned vbits[TM], need[TM];
    {
        const unsigned hw = (unsigned)(Himg * Wimg);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            unsigned qq, r, y, x;
            divmod_magic((unsigned)(m0 + wm * 64 + i * 32 + l31), hw, magic_hw, qq, r);
            divmod_magic(r, (unsigned)Wimg, magic_w, y, x);
            // a tap is outside the image iff it leaves through the pixel's top / bottom / left / right edge: four per-lane flags
            // and four ballots instead of nine bounds tests and ballots (this prologue ran 1.1 us on a lone wave)
            const bool top = y == 0, bot = (int)y == Himg - 1, lef = x == 0, rig = (int)x == Wimg - 1;
            const unsigned row_ok[3] = {top ? 0u : 7u, 7u, bot ? 0u : 7u};                  // taps 3 dy' .. 3 dy' + 2
            const unsigned col_ok = (lef ? 0u : 0x49u) | 0x92u | (rig ? 0u : 0x124u);        // bit tap: dx' = tap % 3
            vbits[i] = (row_ok[0] | row_ok[1] << 3 | row_ok[2] << 6) & col_ok;
            const unsigned at = __builtin_amdgcn_ballot_w64(top) != 0, ab = __builtin_amdgcn_ballot_w64(bot) != 0;
            const unsigned al = __builtin_amdgcn_ballot_w64(lef) != 0, ar = __builtin_amdgcn_ballot_w64(rig) != 0;
            need[i] = (at ? 7u : 0u) | (ab ? 7u << 6 : 0u) | (al ? 0x49u : 0u) | (ar ? 0x124u : 0u);
        }
    }
    unsigned a_addr[TM], w_addr[TN][2];                     // LDS byte addresses: pixel rows in the patch; weight rows per 16-deep chunk
#pragma unroll
    for (int i = 0; i < TM; ++i) a_addr[i] = lds_patch + (unsigned)(((wm * 64 + i * 32 + l31) * RS + h * EPC) * (int)sizeof(T));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
            w_addr[j][kc] = lds_ring + (unsigned)(((wn * WN + j * 32 + l31) * KB + ((2 * kc + h) ^ ((l31 >> 2) & 3)) * EPC) * (int)sizeof(T));

    f32x16 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    // ---- k-loop, software-pipelined over 16-deep chunks (two per stage). Chunk (kt, 0) multiplies out of one fragment buffer
    // while the reads of (kt, 1) fill the other; chunk (kt, 1) does the same with the reads of (kt + 1, 0) - so the stage barrier
    // sits in FRONT of the second chunk of a stage: by then every wave holds the last fragments of stage kt in registers
    // (lgkmcnt(0): nothing of this wave is still reading the buffer the others are about to refill - without that wait the
    // refill overtook reads that were queued behind an LDS-heavy neighbour kernel of another stream, tools/conv_stress.py), the
    // buffer of stage kt is free for stage kt + NST, and stage kt + 1, whose first chunk is read next, must have landed. The
    // pixel fragments come from the resident patch and depend on no barrier at all.
#ifdef LWDETR_CONV_TIMING
    const unsigned long long ct2 = CONV_NOW();
#endif
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                     // the patch and the first NST stages, everybody's
#ifdef LWDETR_CONV_TIMING
    const unsigned long long ct3 = CONV_NOW();
#endif
    // reads of chunk kc of the stage in ring buffer `soff` (byte offset), pixel rows at a_tap[] + (c KB + 16 kc) elements
    auto load = [&](auto c_tag, auto kc_tag, const unsigned (&a_tap)[TM], unsigned soff, V8 (&xf)[TM], V8 (&wf)[TN]) {
        constexpr int OFF = (decltype(c_tag)::value * KB + decltype(kc_tag)::value * 16) * (int)sizeof(T), KC = decltype(kc_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (LWDETR_CONV_ABL & 16) { for (int e = 0; e < 8; ++e) wf[j][e] = (T)(float)(j + e); asm volatile("" : "+v"(wf[j])); continue; }
            lds_read16(wf[j], w_addr[j][KC] + soff);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (LWDETR_CONV_ABL & 16) { for (int e = 0; e < 8; ++e) xf[i][e] = (T)(float)(i - e); asm volatile("" : "+v"(xf[i])); continue; }
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[i]) : "v"(a_tap[i]), "n"(OFF));
        }
    };
    // the fragments of a chunk are in their registers once at most `pending` younger reads are outstanding; the empty statements
    // tie the MFMAs (and the masks) below to the wait
    auto landed = [&](auto pending, V8 (&xf)[TM], V8 (&wf)[TN]) {
        constexpr int P = decltype(pending)::value;
        if (!(LWDETR_CONV_ABL & 16)) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(P) : "memory");
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(wf[j]));
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(xf[i]));
    };
    auto mma = [&](const unsigned (&nd)[TM], const unsigned (&keep)[TM], V8 (&xf)[TM], V8 (&wf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            if (nd[i]) {                              // wave-uniform: some lane's tap falls outside its image
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                u32x4v u = __builtin_bit_cast(u32x4v, xf[i]);
                u = u & keep[i];
                xf[i] = __builtin_bit_cast(V8, u);
            }
        if (LWDETR_CONV_ABL & 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" :: "v"(wf[j]));
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" :: "v"(xf[i]));
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = Mma32<T>::k16(wf[j], xf[i], acc[j][i]);
    };
    auto tap_rows = [&](int tap, unsigned (&a_tap)[TM]) {
        const unsigned toff = (unsigned)((Wimg + 1 + (tap / 3 - 1) * Wimg + (tap % 3 - 1)) * RS * (int)sizeof(T));
#pragma unroll
        for (int i = 0; i < TM; ++i) a_tap[i] = a_addr[i] + toff;
    };
    constexpr unsigned SLOT_B = BN * KB * sizeof(T);
    V8 xa[TM], wa[TN], xb[TM], wb[TN];
    unsigned a_cur[TM], a_nxt[TM];
    tap_rows(0, a_cur);
    load(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, a_cur, 0u, xa, wa);
    int kt = 0;
    unsigned soff = 0;                                // ring buffer of stage kt
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        tap_rows(tap + 1, a_nxt);                     // (tap 9 is never read)
        unsigned nd[TM], keep[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { nd[i] = (need[i] >> tap) & 1; keep[i] = (vbits[i] >> tap) & 1 ? 0xffffffffu : 0u; }
        auto one_stage = [&](auto c_tag) {
            constexpr int C = decltype(c_tag)::value;
            load(c_tag, std::integral_constant<int, 1>{}, a_cur, soff, xb, wb);
            landed(std::integral_constant<int, NRD>{}, xa, wa);
            mma(nd, keep, xa, wa);
#ifdef LWDETR_CONV_TIMING
            const unsigned long long cta = CONV_NOW();
#endif
            landed(std::integral_constant<int, 0>{}, xb, wb);  // ... and no read of stage kt is in flight any more
#ifdef LWDETR_CONV_TIMING
            const unsigned long long ctb = CONV_NOW();
#endif
            wait_vmcnt<(NST - 2) * B_MY>();           // stage kt + 1 has landed (this wave's pieces) ...
#ifdef LWDETR_CONV_TIMING
            const unsigned long long ctc = CONV_NOW();
#endif
            __builtin_amdgcn_s_barrier();             // ... and everybody's; everybody holds the last fragments of stage kt
#ifdef LWDETR_CONV_TIMING
            ct_lgkm += ctb - cta; ct_vm += ctc - ctb; ct_bar += CONV_NOW() - ctc;
#endif
            {
                T* Bs = (T*)((unsigned char*)ring + soff);
#pragma unroll
                for (int k = 0; k < B_MY; ++k) {
                    if (LWDETR_CONV_ABL & 8) continue;
                    gdma16((kt + NST < nk && w_src[k]) ? w_src[k] + (kt + NST) * KB : zero, Bs + (wave + 4 * k) * 64 * EPC);
                }
            }
            soff = soff + SLOT_B == NST * SLOT_B ? 0u : soff + SLOT_B;
            ++kt;
            // first chunk of the next stage: the next tap's rows after the last stage of this one
            if constexpr (C + 1 < NCH) load(std::integral_constant<int, C + 1>{}, std::integral_constant<int, 0>{}, a_cur, soff, xa, wa);
            else if (tap < 8) load(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, a_nxt, soff, xa, wa);
            mma(nd, keep, xb, wb);
        };
        one_stage(std::integral_constant<int, 0>{});
        one_stage(std::integral_constant<int, 1>{});
        one_stage(std::integral_constant<int, 2>{});
        one_stage(std::integral_constant<int, 3>{});
        if constexpr (NCH > 4) { one_stage(std::integral_constant<int, 4>{}); one_stage(std::integral_constant<int, 5>{}); }
#pragma unroll
        for (int i = 0; i < TM; ++i) a_cur[i] = a_nxt[i];
    }
#ifdef LWDETR_CONV_TIMING
    const unsigned long long ct4 = CONV_NOW();
#endif
    wait_vmcnt<0>();            // the dummy tail pieces
    __syncthreads();            // ... and everybody's last fragment reads, before the patch becomes the epilogue's staging area
    // ---- epilogue: 64 rows per pass through the f32 stage area (aliases the patch), finished by all 256 threads.
    // 32x32 accumulator: register 4 q + r of lane (c = lane & 31, hi) is element (row 8 q + 4 hi + r, column c) of D = W x pixels.
    float* stg = (float*)conv_smem;
    constexpr int SLD = BN + 4;
    if (LWDETR_CONV_ABL & 4) {
        float keepalive = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) keepalive += acc[j][i][e];
        if (keepalive == 123.456f) stg[0] = 1.f;
        return;
    }
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (wm == pass) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]};
                        *(f32x4*)(stg + (i * 32 + l31) * SLD + wn * WN + j * 32 + q * 8 + h * 4) = v;
                    }
        }
        __syncthreads();
        epilogue_finish<T, BN, 256>(d, d.seg[0], false, stg, m0 + pass * 64, 0);
        __syncthreads();
    }
#ifdef LWDETR_CONV_TIMING
    if (wg == LWDETR_CONV_TIMING && lane == 0) {
        unsigned long long* o = g_conv_timing[wave];
        const unsigned long long ct5 = CONV_NOW();
        o[0] = ct1 - ct0; o[1] = ct2 - ct1; o[2] = ct3 - ct2; o[3] = ct4 - ct3; o[4] = ct5 - ct4; o[5] = ct_lgkm; o[6] = ct_vm; o[7] = ct_bar; o[8] = ct5 - ct0;
    }
#endif
}

// ---- A-panel-resident variant (16-bit, plain A, K <= 768): one workgroup owns BM rows and ALL N columns. The BM x K panel
// of A is DMA'd into LDS once and stays; the weight matrix streams through an NST-deep ring of 64 x 64 stages while the
// workgroup walks the column tiles. Why: these GEMMs have short K and are bound by what a CU can pull in per output tile -
// a 64 x 64 tile re-fetches its A rows for every column tile (N / 64 times) and its W rows for every row tile; here A
// enters a CU once and W once per BM rows (QKV of block 0, M = 51200, N = 576, K = 192: 108 MB of CU ingress instead of
// 345 MB). LDS image: K / 64 column blocks of [BM][64], each swizzled exactly like a KB = 64 stage of gemm_dma_kernel.
// The DMA goes through inline assembly (untracked by hipcc, see mlp.hip) and is drained by counted waits; the epilogue's
// global stores share vmcnt with it and loads / stores may retire out of order, so the first wait after an epilogue drains
// the counter completely (the ring pieces in flight have had the whole epilogue to land).



// ---- large-tile variant for the compute-bound GEMMs (C = 384 / 768 models: K >= 384, tens of thousands of rows) -----
// 512 threads = 8 waves, block tile 256 x BN (BN = 256: 2 x 4 waves of 128 x 64; BN = 128: 4 x 2 waves of 64 x 64),
// 32x32x16 MFMAs (a 1 KB operand fragment feeds 16384 MACs: half the LDS reads per FLOP of the 16x16x32 form at these
// wave tiles), operands through an NST-deep DMA ring of KB-deep stages exactly like gemm_dma_kernel (counted vmcnt, one raw
// barrier per stage, source-side XOR swizzle). Why a second kernel: a 64 x 64 tile pulls (64 + 64) x K x 2 bytes into the CU
// for 64 x 64 x K MACs - 4x the ingress per FLOP of a 256 x 256 tile - and these shapes are bound by exactly that (C = 768:
// 375 TFLOP/s with 64 x 64, 430 with 128 x 128). LDS swizzle for the 32-row fragments (lane -> row = lane & 31, 16-byte
// k-slot 2 kc + (lane >> 5)): slot ^= (row >> 2) & 3 at KB = 32, (row >> 1) & 7 at KB = 64 - conflict-free for the 16-lane
// ds_read_b128 service groups (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}).
// Ablation builds (-DLWDETR_BIG_VARIANT=bits, timing only - results are wrong with 8 / 32): 4 = no sched_barrier pins in the
// loop, 8 = no DMA inside the loop, 32 = no MFMAs (profiles/r2g_gemm_big_pipeline.txt). The product is built with 0.
#ifndef LWDETR_BIG_VARIANT
#define LWDETR_BIG_VARIANT 0
#endif
// Phase timing for kernel tuning (tools/big_timing.py builds a private copy with -DLWDETR_BIG_TIMING=<workgroup>; never in
// the product): per wave of that workgroup, 10 ns ticks spent in the vmcnt wait, the stage barrier, the MFMA chunks, the
// epilogue, and the whole kernel.
#ifdef LWDETR_BIG_TIMING
__device__ unsigned long long g_big_timing[8][8];
extern "C" int lwdetr_debug_big_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_big_timing), sizeof(g_big_timing)) == hipSuccess ? 0 : 1;
}
#define BIG_NOW() __builtin_amdgcn_s_memrealtime()
#endif
// Round 5: the same kernel with NW = 4 waves on a BM = 128 row tile (wave tiles unchanged: 128 x 64 at BN = 256, 64 x 96 at BN = 192),
// 32-deep stages in a 3-deep ring = 72 / 60 KB of LDS and <= 256 registers: TWO workgroups per CU. A K = 768 tile of the 8-wave
// form spends a third of its time in its prologue (first stages on their way from HBM, every workgroup of the chip at once) and its
// epilogue (256 KB through LDS, 128 KB of stores) with the CU's matrix pipes idle; with a second, independent workgroup on the CU
// one tile's epilogue / prologue runs beside the other's k-loop (profiles/r5a_*).
template <typename T, int BN, int KB, int NST, int AMODE, int BM, int NW, bool LN = false>
__device__ __forceinline__ void gemm_big_body(const lwdetr_gemm_desc& d) {
    static_assert(sizeof(T) == 2, "16-bit types only");
    static_assert((NW == 8 && BM == 256) || (NW == 4 && BM == 128), "8 waves x 256 rows or 4 waves x 128 rows");
    constexpr int EPC = 8;
    constexpr int WGN = BN == 256 ? 4 : 2, WGM = NW / WGN;        // wave grid: 2 x 4 (BN 256) or 4 x 2 (BN 192 / 128); NW = 4: 1 x 4 / 2 x 2
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static_assert(WN % 32 == 0 && WM % 64 == 0 && (BN / (64 / (KB / 8))) % NW == 0 && (BM / (64 / (KB / 8))) % NW == 0, "tile / stage combination not supported");
    constexpr int SLOTS = KB / EPC, RP = 64 / SLOTS, KC = KB / 16;
    constexpr int A_MY = BM / RP / NW, B_MY = BN / RP / NW, PER_STAGE = A_MY + B_MY;
    constexpr int STAGE = (BM + BN) * KB;                         // elements
    typedef typename Vec<T>::v8 V8;
    extern __shared__ __attribute__((aligned(16))) char big_smem[];
    T* smem = (T*)big_smem;

    const int tiles_n = (d.N + BN - 1) / BN;
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm_ = wg / tiles_n, tn_ = wg - tm_ * tiles_n;
    const long m0 = (long)tm_ * BM;
    const int n0 = tn_ * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int wm = wave / WGN, wn = wave % WGN;
    const T* __restrict__ A = (const T*)d.A;
    const T* __restrict__ W = (const T*)d.W;
    const T* zero = (const T*)g_zero16;

    // DMA pieces of this wave (piece i covers tile rows RP i .. RP i + RP - 1; lane -> row i RP + lane / SLOTS, physical slot
    // lane % SLOTS; the logical 16-byte column it fetches is the slot un-swizzled with the row's key). Pieces 0 .. A_MY - 1
    // are A rows, the rest W rows; rows past M / N read the zero page (step 0). The DMA goes through inline assembly
    // (gdma16: invisible to hipcc, so pending pieces do not turn its ds_read waits into lgkmcnt(0)) and the pieces of a
    // stage are issued two at a time BETWEEN the MFMA groups of the previous stage: issued in one burst after the barrier
    // they cost ~100 issue cycles each with both waves of a SIMD stalled behind them (kb64: 8 pieces per wave and stage).
    const int prow = lane / SLOTS, pslot = lane % SLOTS;
    auto key_of = [](int row) { return KB == 32 ? (row >> 2) & 3 : (row >> 1) & 7; };
    // Piece kk of an operand covers tile rows RP (wave + NW kk) + prow: NW RP rows further per piece, and the swizzle key of
    // those rows does not depend on kk (NW RP is a multiple of the key period) - so a lane keeps ONE base offset per operand
    // and adds the wave-uniform piece / stage displacement when it issues (the round-1 kernel kept a pointer and a step per
    // piece: 24 registers the accumulators need).
    const int row0 = RP * wave + prow;
    const int swz = (pslot ^ key_of(row0)) * EPC;
    static_assert((NW * RP) % 16 == 0, "piece stride must preserve the swizzle key");
    const long a_row0 = m0 + row0, w_row0 = (long)n0 + row0;
    const long a_off0 = a_row0 * d.lda + swz, w_off0 = w_row0 * (long)d.K + swz;
    // implicit-GEMM 3x3 view (AMODE CONV3x3): A row = output pixel (b, y, x); stage k0 reads channels k0 % Cin .. of the input
    // pixel (y s + tap / 3 - 1, x s + tap % 3 - 1), tap = k0 / Cin (zero page outside the image). The shifted source row
    // only changes with the tap - every Cin / KB stages - so its address is cached per piece (gemm_dma_kernel does the same);
    // the pixel coordinates are packed into one register per piece (b: 12 bits, y: 10, x: 10; -1 = row past M).
    int cv_pix[A_MY]; const T* cv_src[A_MY];
    if (AMODE == LWDETR_A_CONV3x3) {
#pragma unroll
        for (int k = 0; k < A_MY; ++k) {
            const long gr = a_row0 + NW * RP * k;
            const int hw = d.conv_hout * d.conv_wout;
            const int b = (int)(gr / hw), r = (int)(gr - (long)b * hw), y = r / d.conv_wout, x = r - y * d.conv_wout;
            cv_pix[k] = gr < d.M ? (b << 20) | (y << 10) | x : -1;
            cv_src[k] = nullptr;
        }
    }
    const int nk = d.K / KB;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)big_smem;
    const unsigned wave_off = (unsigned)__builtin_amdgcn_readfirstlane(wave * 64 * EPC * (int)sizeof(T));
    // issue piece k of stage kt (stages past the end of K read the zero page: the wait counts stay compile-time constants)
    auto issue_piece = [&](int kt, int k) {
        const bool is_a = k < A_MY;
        const unsigned dst = lds0 + (unsigned)((kt % NST) * STAGE + (is_a ? 0 : BM * KB)) * (unsigned)sizeof(T) + wave_off +
                             (unsigned)(NW * (is_a ? k : k - A_MY) * 64 * EPC * (int)sizeof(T));
        const int kk = is_a ? k : k - A_MY;
        const T* src;
        if (AMODE == LWDETR_A_CONV3x3 && is_a) {
            const int k0 = kt * KB, tap = k0 / d.conv_cin, c0 = k0 - tap * d.conv_cin;      // wave-uniform
            if (c0 == 0) {                 // first stage of a tap (stages reach a piece in order): new source pixel
                const int pix = cv_pix[kk], b = pix >> 20, y = (pix >> 10) & 1023, x = pix & 1023;
                const int iy = y * d.conv_stride + tap / 3 - 1, ix = x * d.conv_stride + tap % 3 - 1;
                cv_src[kk] = (pix >= 0 && iy >= 0 && iy < d.a_tok.Hp && ix >= 0 && ix < d.a_tok.Wp)
                                 ? A + tok_encode(b, iy, ix, d.a_tok) * d.lda + d.a_col0 + swz : nullptr;
            }
            src = (kt < nk && cv_src[kk]) ? cv_src[kk] + c0 : zero;
        } else {
            const long disp = (long)(NW * RP * kk) * (is_a ? (long)d.lda : (long)d.K) + (long)kt * KB;       // wave-uniform
            const bool ok = kt < nk && (is_a ? a_row0 + NW * RP * kk < d.M : w_row0 + NW * RP * kk < (long)d.N);
            src = ok ? (is_a ? A + a_off0 : W + w_off0) + disp : zero;
        }
        const unsigned m0v = __builtin_amdgcn_readfirstlane(dst);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(src) : "memory");
    };

    int si = 0;
#pragma unroll
    for (int s = 1; s < 3; ++s) if (s < d.nseg && n0 >= d.seg[s].n_begin) si = s;
    const lwdetr_gemm_seg& sg = d.seg[si];
    const bool col_orient = sg.mode == LWDETR_OUT_HEADS_T;

    int pofs[KC];                            // element offset of this lane's k-run inside its row, per 16-deep chunk
#pragma unroll
    for (int c = 0; c < KC; ++c) pofs[c] = ((2 * c + h) ^ key_of(m)) * EPC;
    const int arow = (wm * WM + m) * KB, brow = (wn * WN + m) * KB;
    float* stg = (float*)big_smem;
    constexpr int SLD = BN + 4, SLD_T = 64 + 4;

    // The whole k-loop + accumulator staging is instantiated once per MFMA operand order (a branch on the orientation
    // inside the loop made hipcc keep both orders alive and spill the 128 accumulator registers around every MFMA group).
    auto body = [&](auto col_tag) {
        constexpr bool COL = decltype(col_tag)::value;
        f32x16 acc[TN][TM];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
        // Software pipeline over (stage, 16-deep chunk) slots. Slot (kt, c) multiplies chunk c of stage kt out of one fragment
        // buffer while it reads the NEXT chunk - (kt, c + 1), or (kt + 1, 0) in the last slot - into the other one, one
        // ds_read after each MFMA (the LDS queue sees a steady trickle instead of 8 waves x 6 reads right after a barrier, and
        // no chunk starts on fragments that were requested just before it). The stage barrier sits in FRONT of the last slot
        // of a stage: by then every wave holds the last fragments of stage kt in registers, so the buffer of stage kt is free
        // for the A pieces of stage kt + NST (issued in that slot; the W pieces of a stage follow one slot later), and stage
        // kt + 1 - whose first chunk the slot reads - must have landed: A pieces have had a whole step, W pieces KC - 1 slots.
#pragma unroll
        for (int s_ = 0; s_ < NST - 1; ++s_)
#pragma unroll
            for (int k = 0; k < PER_STAGE; ++k) issue_piece(s_, k);
#pragma unroll
        for (int k = 0; k < A_MY; ++k) issue_piece(NST - 1, k);
#ifdef LWDETR_BIG_TIMING
        unsigned long long tt_wait = 0, tt_bar = 0, tt_mma = 0;
        const unsigned long long tt_start = BIG_NOW();
#endif
        V8 xf[2][TM], wf[2][TN];
        wait_vmcnt<(NST - 2) * PER_STAGE + A_MY>();        // stage 0 has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                      // ... and everybody's
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[0][i] = *(const V8*)(smem + arow + i * 32 * KB + pofs[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[0][j] = *(const V8*)(smem + BM * KB + brow + j * 32 * KB + pofs[0]);
        for (int kt = 0; kt < nk; ++kt) {
            const T* As = smem + (kt % NST) * STAGE;
            const T* An = smem + ((kt + 1) % NST) * STAGE;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                constexpr int NM = TN * TM, NR = TM + TN;
                const bool last = c == KC - 1;
                if (last) {
#ifdef LWDETR_BIG_TIMING
                    const unsigned long long ta = BIG_NOW();
#endif
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the last fragments of stage kt are in registers
                    wait_vmcnt<(NST - 2) * PER_STAGE>();                 // stage kt + 1 has landed (this wave's pieces)
#ifdef LWDETR_BIG_TIMING
                    const unsigned long long tb = BIG_NOW();
#endif
                    __builtin_amdgcn_s_barrier();
#ifdef LWDETR_BIG_TIMING
                    tt_wait += tb - ta; tt_bar += BIG_NOW() - tb;
#endif
                }
                const T* Ar = last ? An : As;
                const T* Br = Ar + BM * KB;
                const int po = pofs[(c + 1) % KC];
                const int nb = (c + 1) & 1, cb = c & 1;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int idx = 0; idx < (NM > NR ? NM : NR); ++idx) {
                    if (idx < NM) {
                        const int j = idx / TM, i = idx % TM;
#if LWDETR_BIG_VARIANT & 32       // ablation (timing only): no MFMAs
                        asm volatile("" : "+v"(acc[j][i]) : "v"(xf[cb][i]), "v"(wf[cb][j]));
#else
                        acc[j][i] = COL ? Mma32<T>::k16(xf[cb][i], wf[cb][j], acc[j][i]) : Mma32<T>::k16(wf[cb][j], xf[cb][i], acc[j][i]);
#endif
                    }
                    // one fragment read after each MFMA, in the order the next slot's MFMAs want them: x0, w0, x1 .. x(TM-1),
                    // w1 .. (two or three per MFMA - all reads out by the third MFMA - measured 3-4 % slower: bursts on the LDS queue)
                    if (idx == 0) xf[nb][0] = *(const V8*)(Ar + arow + po);
                    else if (idx == 1) wf[nb][0] = *(const V8*)(Br + brow + po);
                    else if (idx < TM + 1) xf[nb][idx - 1] = *(const V8*)(Ar + arow + (idx - 1) * 32 * KB + po);
                    else if (idx < NR) wf[nb][idx - TM] = *(const V8*)(Br + brow + (idx - TM) * 32 * KB + po);
#if !(LWDETR_BIG_VARIANT & 8)         // ablation (timing only): 8 = no DMA inside the loop
                    if (last && idx < A_MY) issue_piece(kt + NST, idx);                  // A pieces of stage kt + NST
                    if (c == 0 && idx < B_MY) issue_piece(kt + NST - 1, A_MY + idx);     // W pieces of stage kt + NST - 1
#endif
#if !(LWDETR_BIG_VARIANT & 4)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
                static_assert(A_MY <= (NM > NR ? NM : NR) && B_MY <= (NM > NR ? NM : NR), "one DMA piece per MFMA at most");
            }
        }
#ifdef LWDETR_BIG_TIMING
        const unsigned long long tt_loop = BIG_NOW();
        tt_mma = tt_loop - tt_start - tt_wait - tt_bar;
#endif
        wait_vmcnt<0>();        // the dummy tail pieces (hipcc does not know about them)
        __syncthreads();        // drains the dummy tail pieces and the last fragment reads before LDS is re-used
#ifdef LWDETR_BIG_TIMING
        unsigned long long te_stage = 0, te_fin = 0, te_sync = 0; const unsigned long long te_drain = BIG_NOW();
#endif
        // ---- epilogue: 64-row blocks through the f32 stage area (aliases the ring), finished by all 512 threads. A pass
        // stages EPI_SLOTS blocks at once - chosen so that every wave has tiles in every pass (block b of the tile belongs
        // to pass b % EPI_PASSES, stage slot b / EPI_PASSES) - and is bracketed by one pair of workgroup barriers: two
        // passes for BN = 256 / 192, one for BN = 128 (round 1: four passes of one block, half or a quarter of the waves
        // staging in each: 9-12 us per tile, a third of a K = 768 tile's time; a register-direct epilogue - permlane32
        // swaps to 16-byte runs, no LDS - measured slower, 14-19 us: its stores touch 32 rows x 32 bytes per instruction).
        // 32x32 accumulator: register 4 q + r of lane (c = lane & 31, hi) is element (row 8 q + 4 hi + r, column c) of D.
        constexpr int EPI_PASSES = BN == 128 ? 1 : 2, EPI_SLOTS = (BM / 64) / EPI_PASSES;
#pragma unroll 1
        for (int pass = 0; pass < EPI_PASSES; ++pass) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int blk = wm * (WM / 64) + (i >> 1);
                if (blk % EPI_PASSES != pass) continue;                 // wave-uniform: this tile's rows belong to the pass
                const int ii = i & 1;
                float* sb = stg + (blk / EPI_PASSES) * (COL ? BN * SLD_T : 64 * SLD);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]};
                        if (!COL)    // lane: row ii * 32 + (lane & 31), columns wn WN + 32 j + 8 q + 4 hi .. + 3
                            *(f32x4*)(sb + (ii * 32 + m) * SLD + wn * WN + j * 32 + q * 8 + h * 4) = v;
                        else         // lane: column wn WN + 32 j + (lane & 31), rows ii * 32 + 8 q + 4 hi .. + 3
                            *(f32x4*)(sb + (wn * WN + j * 32 + m) * SLD_T + ii * 32 + q * 8 + h * 4) = v;
                    }
            }
#ifdef LWDETR_BIG_TIMING
            const unsigned long long t0_ = BIG_NOW();
#endif
            __syncthreads();
#ifdef LWDETR_BIG_TIMING
            const unsigned long long t1_ = BIG_NOW();
#endif
#pragma unroll 1
            for (int slot = 0; slot < EPI_SLOTS; ++slot)       // not unrolled: the accumulators of the later passes are still live
                if constexpr (LN) epilogue_finish_ln<T, BN, NW * 64, 4>(d, sg, COL, stg + slot * (COL ? BN * SLD_T : 64 * SLD), m0 + (slot * EPI_PASSES + pass) * 64, n0);
                else epilogue_finish<T, BN, NW * 64, 4>(d, sg, COL, stg + slot * (COL ? BN * SLD_T : 64 * SLD), m0 + (slot * EPI_PASSES + pass) * 64, n0);
#ifdef LWDETR_BIG_TIMING
            const unsigned long long t2_ = BIG_NOW();
#endif
            __syncthreads();
#ifdef LWDETR_BIG_TIMING
            te_stage += t1_ - t0_; te_fin += t2_ - t1_; te_sync += BIG_NOW() - t2_;
#endif
        }
#ifdef LWDETR_BIG_TIMING
        if (blockIdx.x == LWDETR_BIG_TIMING && lane == 0) {
            const unsigned long long te = BIG_NOW();
            unsigned long long* o = g_big_timing[wave];
            o[0] = tt_wait; o[1] = tt_bar; o[2] = tt_mma; o[3] = te - tt_loop; o[4] = te - tt_start; o[5] = nk;
            o[6] = (te_drain - tt_loop) | (te_stage << 16) | (te_fin << 32) | (te_sync << 48);     // 10 ns ticks, 16 bits each
        }
#endif
    };
    if (col_orient) body(std::true_type{});
    else body(std::false_type{});
}
// Two entry points, because the launch bounds differ: the 8-wave form keeps round 2's plain __launch_bounds__(512) - with a second argument
// (amdgpu_waves_per_eu) hipcc schedules the SAME body differently and the QKV launch of xlarge (HEADS / HEADS_T epilogues) ran 20 % slower
// (260 -> 312 us, same box: profiles/r5g_*) - and the 4-wave form asks for two workgroups per CU.
template <typename T, int BN, int KB, int NST, int AMODE = LWDETR_A_PLAIN>
__global__ __launch_bounds__(512) void gemm_big_kernel(const lwdetr_gemm_desc d) { gemm_big_body<T, BN, KB, NST, AMODE, 256, 8>(d); }
// Round 6: the forms that were built, validated and MEASURED SLOWER (or worth nothing) - the 4-wave / 128-row form (two workgroups per CU, r5a),
// the LayerNorm-folded epilogue (r5d, r5g), split-K of the 64 x 64 ring kernel (r5e), and attention.hip's LDS window-tile kernel (r5c) - are
// compiled only with `make TUNE=-DLWDETR_EXPERIMENTS` (their tests skip otherwise; lwdetr_has_experiments()). The default library carries the
// kernels of the launch plan only: every resident variant was a standing risk of moving the product kernels' register allocation (r5g).
#ifdef LWDETR_EXPERIMENTS
template <typename T, int BN, int KB, int NST, int AMODE = LWDETR_A_PLAIN>
__global__ __launch_bounds__(256, 2) void gemm_big4_kernel(const lwdetr_gemm_desc d) { gemm_big_body<T, BN, KB, NST, AMODE, 128, 4>(d); }
// the 256 x 256 tile with the LayerNorm-folded epilogue (every segment of the launch carries ln_stats): its own kernel, see epilogue_finish_ln
template <typename T>
__global__ __launch_bounds__(512) void gemm_big_ln_kernel(const lwdetr_gemm_desc d) { gemm_big_body<T, 256, 64, 2, LWDETR_A_PLAIN, 256, 8, true>(d); }
#endif

template <typename T, int BN, int KB, int NST, int AMODE = LWDETR_A_PLAIN, int BM = 256, int NW = 8, bool LN = false>
int launch_big(const lwdetr_gemm_desc& d, hipStream_t st) {
    static_assert(!LN || (BN == 256 && KB == 64 && NST == 2 && AMODE == LWDETR_A_PLAIN && NW == 8), "the LayerNorm-folded epilogue exists for the 256 x 256 tile");
    constexpr size_t ring = (size_t)NST * (BM + BN) * KB * sizeof(T);
    constexpr size_t stg = (size_t)((BM / 64) / (BN == 128 ? 1 : 2)) * (64 * (BN + 4) > BN * 68 ? 64 * (BN + 4) : BN * 68) * sizeof(float);
    constexpr size_t lds = ring > stg ? ring : stg;
    // per device (a process may drive several GPUs); a device that refuses the 160 KB request keeps the 64 x 64 ring kernel:
    // LWDETR_ERR_UNSUPPORTED tells try_launch_big to fall through
    static signed char state[16] = {};          // 0 = not asked yet, 1 = granted, -1 = refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_UNSUPPORTED;
    if (state[dev] == 0)
    {
        const void* fn;
#ifdef LWDETR_EXPERIMENTS
        if constexpr (LN) fn = (const void*)gemm_big_ln_kernel<T>;
        else if constexpr (NW == 4) fn = (const void*)gemm_big4_kernel<T, BN, KB, NST, AMODE>;
        else
#endif
        fn = (const void*)gemm_big_kernel<T, BN, KB, NST, AMODE>;
        state[dev] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, NW == 4 ? 80 * 1024 : 160 * 1024) == hipSuccess ? 1 : -1;
    }
    if (state[dev] < 0) { (void)hipGetLastError(); return LWDETR_ERR_UNSUPPORTED; }
    static_assert(NW == 8 || lds <= 80 * 1024, "two workgroups per CU");
    const long nwg = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
#ifdef LWDETR_EXPERIMENTS
    if constexpr (LN) hipLaunchKernelGGL((gemm_big_ln_kernel<T>), dim3((unsigned)nwg), dim3(512), lds, st, d);
    else if constexpr (NW == 4) hipLaunchKernelGGL((gemm_big4_kernel<T, BN, KB, NST, AMODE>), dim3((unsigned)nwg), dim3(256), lds, st, d);
    else
#endif
    hipLaunchKernelGGL((gemm_big_kernel<T, BN, KB, NST, AMODE>), dim3((unsigned)nwg), dim3(512), lds, st, d);
    return lwdetr_check_launch();
}

// Shapes the large-tile kernel takes: plain A or the implicit-GEMM 3x3 view, no A2, K (and Cin) a multiple of 64, segment
// boundaries on the column tile, and (mode 1, the default) enough work for 256-row tiles. Measured crossover (tools/op_times.py --gemm-big 2, M = 9600 .. 217600): K >= 384
// wins from M = 16384 rows; with K >= 960 already from ~100 tiles (M = 9600 .. 12800, N = 384); short-K / few-tile shapes
// (M = 9600: N 256 K 256 11 -> 25 us, N 2048 K 256 35 -> 49, N 256 K 2048 33 -> 70 on 38 tiles) stay on the 64 x 64 ring.
// LWDETR_GEMM_BIG / lwdetr_gemm_tuning(): 0 = never, 1 = default thresholds, 2 = whenever legal (tests), 32 / 64 = whenever
// legal with that stage depth (tuning).
int g_big_mode = -1;
// Shapes on which the 4-wave / 128-row form (two workgroups per CU) measured faster than the 8-wave 256-row one: none by a margin
// that shows at model level (profiles/r5a_gemm_big_2wg_and_start_skew.txt: xlarge K = 768 shapes -3 .. +3 %, K = 3072 -23 %, C = 384
// shapes -2 .. +6 %; xlarge 960x960 786 -> 750 img/s, large +-0, medium +1.4 %). A 128 x 256 tile pulls 1.5x the bytes per MFMA into
// the CU and needs 32-deep stages (a barrier and six DMA issues per 16 MFMAs of a wave) to fit twice: what the second workgroup hides of
// the other's prologue and epilogue the slower k-loop gives back. The form stays selectable (LWDETR_GEMM_BIG_2WG=2, tests).
template <int AMODE>
bool big_2wg_pays(const lwdetr_gemm_desc& d, int bn) {
    (void)d; (void)bn;
    return false;
}
template <typename T, int AMODE>
int try_launch_big(const lwdetr_gemm_desc& d, hipStream_t st, bool& taken) {
    taken = false;
    if constexpr (sizeof(T) != 2 || AMODE == LWDETR_A_PATCH16) return LWDETR_OK;
    else {
        const int mode = g_big_mode >= 0 ? g_big_mode : (int)lwdetr_knob(KNOB_GEMM_BIG, 1);
        if (!mode || d.A2 || d.K % 64 != 0 || d.N < 128) return LWDETR_OK;
        if (AMODE == LWDETR_A_CONV3x3 && (d.conv_cin % 64 != 0 || d.a_col0 % 8 != 0 || d.conv_hout > 1024 || d.conv_wout > 1024 ||
                                          d.M / ((long)d.conv_hout * d.conv_wout) >= 2048)) return LWDETR_OK;   // packed pixel coordinates
        // column tile: 256 where N and the segment boundaries allow; 192 for N <= 192 (the 3x3 convolutions of the C2f blocks:
        // ONE column tile instead of two half-empty 128-wide ones); else 128. (Measured: N = 384 as 2 x 192 is 5-20 % slower
        // than 3 x 128 - fewer, fatter tiles on a 2-deep ring - and N = 1152 ties.)
        int bn = d.N % 256 == 0 || d.N > 512 ? 256 : (d.N <= 192 ? 192 : 128);
        bn = (int)lwdetr_knob(KNOB_GEMM_BIG_BN, bn);       // tuning: force the column tile where legal
        for (int s = 0; s < d.nseg; ++s)
            if (d.seg[s].n_begin % bn != 0) bn = 128;
        for (int s = 0; s < d.nseg; ++s) if (d.seg[s].n_begin % bn != 0) return LWDETR_OK;
        const long tiles = ((d.M + 255) / 256) * ((d.N + bn - 1) / bn);
        if (mode == 1 && !(d.K >= 384 && d.N >= 192 && (d.M >= 16384 || (d.K >= 960 && tiles >= 96)))) return LWDETR_OK;
        const int variant = mode >= 10 ? mode : 0;     // tuning: 32 / 64 = stage depth (ring 4 / 2 deep), 128 = the 4-wave 128-row form
        // 4-wave / 128-row form (two workgroups per CU): LWDETR_GEMM_BIG_2WG = 0 never, 1 where it measured faster (default), 2 whenever legal
        const int wg2_mode = (int)lwdetr_knob(KNOB_GEMM_BIG_2WG, 1);
        const bool wg2 = variant == 128 || (variant == 0 && (wg2_mode == 2 || (wg2_mode == 1 && big_2wg_pays<AMODE>(d, bn))));
        int rc;
        bool ln = false;
        for (int s_ = 0; s_ < d.nseg; ++s_) ln = ln || d.seg[s_].ln_stats != nullptr;
        if (ln) {       // LayerNorm folded into the GEMM: the 256 x 256 tile's own kernel or nothing (lwdetr_gemm reports the rest as unsupported)
#ifdef LWDETR_EXPERIMENTS
            if constexpr (AMODE == LWDETR_A_PLAIN) {
                if (bn == 256) { rc = launch_big<T, 256, 64, 2, LWDETR_A_PLAIN, 256, 8, true>(d, st); taken = rc != LWDETR_ERR_UNSUPPORTED; return taken ? rc : LWDETR_OK; }
            }
#endif
            return LWDETR_OK;
        }
#ifdef LWDETR_EXPERIMENTS
        if (wg2 && bn == 256) rc = launch_big<T, 256, 32, 3, AMODE, 128, 4>(d, st);
        else if (wg2 && bn == 192) rc = launch_big<T, 192, 32, 3, AMODE, 128, 4>(d, st);
        else
#else
        if (wg2 && bn != 128) return LWDETR_ERR_UNSUPPORTED;       // the 4-wave form was asked for by name: not in this build
#endif
        if (bn == 256) rc = variant == 32 ? launch_big<T, 256, 32, 4, AMODE>(d, st) : launch_big<T, 256, 64, 2, AMODE>(d, st);
        else if (bn == 192) rc = launch_big<T, 192, 64, 2, AMODE>(d, st);
        else rc = variant == 32 ? launch_big<T, 128, 32, 4, AMODE>(d, st) : launch_big<T, 128, 64, 3, AMODE>(d, st);
        taken = rc != LWDETR_ERR_UNSUPPORTED;          // refused LDS size: the caller launches the ring kernel instead
        return taken ? rc : LWDETR_OK;
    }
}

// The patch-resident 3x3 convolution takes: stride 1, raster token rows in and out (the C2f bottleneck convolutions of the
// projector), one LINEAR segment, N = Cin in {128, 192}, and a patch (128 + 2 W + 2 rows) that fits beside the weight ring in
// 160 KB. LWDETR_CONV_PATCH: 0 = off, 1 = default, 2 = every legal shape (tuning / A-B runs / tests).
template <typename T, int CIN, int NST>
int launch_conv_patch(const lwdetr_gemm_desc& d, hipStream_t st) {
    constexpr int SPR = CIN / 8 + 1;
    const int pr = 128 + 2 * d.conv_wout + 2;
    const int np = (pr * SPR + 63) / 64;
    size_t lds = (size_t)np * 1024 + (size_t)NST * CIN * 32 * sizeof(T);
    const size_t stg = (size_t)64 * (CIN + 4) * sizeof(float);
    if (lds < stg) lds = stg;
    if (lds > 160 * 1024) return LWDETR_ERR_UNSUPPORTED;
    static signed char state[16] = {};          // per device: 0 = not asked yet, 1 = granted, -1 = refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_UNSUPPORTED;
    if (state[dev] == 0)
        state[dev] = hipFuncSetAttribute((const void*)conv3x3_patch_kernel<T, CIN, CIN, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess ? 1 : -1;
    if (state[dev] < 0) { (void)hipGetLastError(); return LWDETR_ERR_UNSUPPORTED; }
    const long nwg = (d.M + 127) / 128;
    const unsigned hw = (unsigned)(d.conv_hout * d.conv_wout);
    hipLaunchKernelGGL((conv3x3_patch_kernel<T, CIN, CIN, NST>), dim3((unsigned)nwg), dim3(256), lds, st, d, np,
                       (unsigned)(0x100000000ull / hw), (unsigned)(0x100000000ull / (unsigned)d.conv_wout));
    return lwdetr_check_launch();
}

template <typename T>
int try_launch_conv_patch(const lwdetr_gemm_desc& d, hipStream_t st, bool& taken) {
    taken = false;
    if constexpr (sizeof(T) != 2) return LWDETR_OK;
    else {
        const int mode = (int)lwdetr_knob(KNOB_CONV_PATCH, 1);     // 0 = never, 1 = default (N = 128), 2 = whenever legal
        if (mode == 0) return LWDETR_OK;
        // N = Cin = 192 (the C = 384 models: 117-153 KB of LDS, one workgroup per CU) measured SLOWER than the 256-row large-tile
        // kernel with its 192-wide column tile (large, B = 32: 16 launches 2.24 vs 1.93 ms per step) - off unless asked for
        if (mode == 1 && d.N != 128) return LWDETR_OK;
        // one workgroup per 128 pixels walks all 9 Cin / 32 stages by itself: below ~100 workgroups (a single 640 x 640 image has
        // 13) the 64 x 64 ring kernel's 4x finer grid finishes sooner
        if (mode == 1 && d.M < 100 * 128) return LWDETR_OK;
        const lwdetr_gemm_seg& g = d.seg[0];
        const long hw = (long)d.conv_hout * d.conv_wout;
        if (d.conv_stride != 1 || d.a_tok.winmajor || d.a_tok.Hp != d.conv_hout || d.a_tok.Wp != d.conv_wout || d.nseg != 1 ||
            g.mode != LWDETR_OUT_LINEAR || d.N != d.conv_cin || d.K != 9 * d.conv_cin || (d.N != 128 && d.N != 192) ||
            d.lda % 8 != 0 || d.a_col0 % 8 != 0 || ((size_t)d.A & 15) != 0 || ((size_t)d.W & 15) != 0 || hw <= 1 || d.M % hw != 0 ||
            d.conv_wout < 2 || (long)(d.M + 256) * d.lda >= (1L << 30))       // 32-bit element offsets into A
            return LWDETR_OK;
        // ring depth 3 (two workgroups per CU at N = 128); 4 / 6 / 8 measured no faster (profiles/r3e_conv3x3_patch.txt)
        const int rc = d.N == 128 ? launch_conv_patch<T, 128, 3>(d, st) : launch_conv_patch<T, 192, 3>(d, st);
        taken = rc != LWDETR_ERR_UNSUPPORTED;
        return taken ? rc : LWDETR_OK;
    }
}

template <typename T, int AMODE>
int launch(const lwdetr_gemm_desc& d, hipStream_t st) {
    // column tile: 128 unless a segment boundary (or a small N) asks for 64
    bool bn64 = d.N <= 64 || sizeof(T) == 4;   // f32 (parity mode) tiles are 128 x 64 to stay inside 64 KB of LDS
    for (int s = 0; s < d.nseg; ++s)
        if ((d.seg[s].n_begin % 128) != 0) bn64 = true;
    const int BNsel = bn64 ? 64 : 128;
    for (int s = 0; s < d.nseg; ++s)
        if (d.seg[s].n_begin % BNsel != 0) return LWDETR_ERR_UNSUPPORTED;
    long tiles_m = (d.M + 127) / 128, tiles_n = (d.N + BNsel - 1) / BNsel;
    // small problems (decoder / head GEMMs: a few thousand rows) would leave most of the 256 CUs idle with 128-row
    // tiles: switch to 64 x 64 tiles when the 128-row grid has fewer than ~1.5 workgroups per CU
    // Measured per shape on MI355X (tools/op_times.py with LWDETR_GEMM_TILE=1|2|3, B=32 small model): 64 x 64 tiles win
    // or tie on every plain / patch GEMM of the network (K <= 2048, N <= 2048: short k-loops, the tile's prologue and
    // epilogue latency is what more, smaller workgroups hide); only the implicit-GEMM 3x3 convolutions (K = 9 Cin, a
    // gathered A panel worth re-using across 128 columns) are faster with 128 x 128.
    bool small = AMODE != LWDETR_A_CONV3x3 || tiles_m * tiles_n < 384;
    if (lwdetr_knob_is_set(KNOB_GEMM_TILE)) {                      // tuning: 1 = 64x64, 2 = 128x64, 3 = 128x128 (where legal)
        const int tsel = (int)lwdetr_knob(KNOB_GEMM_TILE, 0);
        if (tsel == 1) small = true;
        if (tsel == 2) { small = false; bn64 = true; tiles_n = (d.N + 63) / 64; }
        if (tsel == 3 && !bn64) small = false;
    }
    if (small) { tiles_m = (d.M + 63) / 64; tiles_n = (d.N + 63) / 64; }
    const long nwg = tiles_m * tiles_n;
    if (nwg <= 0 || nwg > 0x7fffffffL) return LWDETR_ERR_BAD_ARG;
    const int kid = AMODE == LWDETR_A_PLAIN ? KID_GEMM : (AMODE == LWDETR_A_CONV3x3 ? KID_GEMM_CONV : KID_GEMM_PATCH);
    ProfScope ps(kid, 2.0 * d.M * d.N * d.K, ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N) * sizeof(T), st);
    if constexpr (AMODE == LWDETR_A_CONV3x3) {
        bool taken = false;
        const int rc = try_launch_conv_patch<T>(d, st, taken);
        if (taken) return rc;
    }
    if constexpr (AMODE == LWDETR_A_PLAIN && sizeof(T) == 2) {       // round 6: the persistent form of the large-tile kernel (gemm_pt.hip) where it applies
        bool taken = false;
        const int rc = lwdetr_gemm_pt_try(d, std::is_same<T, f16>::value ? DT_F16 : DT_BF16, st, taken);
        if (taken) return rc;
    }
    {
        bool taken = false;
        const int rc = try_launch_big<T, AMODE>(d, st, taken);
        if (taken) return rc;
        for (int s_ = 0; s_ < d.nseg; ++s_)
            if (d.seg[s_].ln_stats) return LWDETR_ERR_UNSUPPORTED;      // the folded LayerNorm exists in the large-tile kernel's epilogue only
    }
    if constexpr (sizeof(T) == 2) {
        const int mode = (int)lwdetr_knob(KNOB_GEMM_DMA, 3);
        if (mode && !d.A2) {
            // 64-deep stages measured equal to 32-deep ones on every GEMM of the network (0.879 vs 0.880 ms per step): the
            // k-loop is not where these short-K GEMMs spend their time. Kept selectable for tuning (LWDETR_GEMM_KB=64).
            const bool kb64 = lwdetr_knob(KNOB_GEMM_KB, 32) == 64 && d.K % 64 == 0 && (AMODE != LWDETR_A_CONV3x3 || d.conv_cin % 64 == 0);
            // (An A-panel-resident schedule - the whole K panel of 64 / 128 rows in LDS, column tiles streamed past it - was built
            // and measured in round 2 and removed in round 3: at M = 51200 QKV (N 576, K 192) 51 -> 73 / 106 us, projector 1x1
            // (N 256, K 768) 60 -> 137 us, value projection (N 768, K 256) 62 -> 93 / 124 us. It cuts what a CU pulls in by 2-3x
            // and loses anyway: with 66-108 KB of LDS a CU holds one or two workgroups whose column tiles, epilogues and store
            // latencies run back to back, where the 64 x 64 grid keeps 4-5 independent workgroups per CU in flight.)
            if (small && kb64) hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 64, AMODE, 3, 64>), dim3((unsigned)nwg), dim3(256), 0, st, d);
            else if (small) {
                // ring depth of the 64 x 64 kernel trades prefetch distance against workgroups per CU (24 KB of LDS at depth 3:
                // six per CU). Measured on the whole network: depth 4 0.894 ms, depth 3 0.847 ms, depth 2 0.868 ms per step.
                const int nst = (int)lwdetr_knob(KNOB_GEMM_NST, 3);
                if (nst == 2) hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 64, AMODE, 2>), dim3((unsigned)nwg), dim3(256), 0, st, d);
#ifdef LWDETR_EXPERIMENTS
                else if (nst == 3 && d.splitk >= 2 && d.splitk_ws && d.K / 32 >= d.splitk)          // round 5: few rows, long K (see gemm_dma_kernel)
                    hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 64, AMODE, 3, 32, true>), dim3((unsigned)(nwg * d.splitk)), dim3(256), 0, st, d);
#endif
                else if (nst == 3) hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 64, AMODE, 3>), dim3((unsigned)nwg), dim3(256), 0, st, d);
                else hipLaunchKernelGGL((gemm_dma_kernel<T, 64, 64, AMODE, 4>), dim3((unsigned)nwg), dim3(256), 0, st, d);
            }
            else if (bn64) hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 64, AMODE, 4>), dim3((unsigned)nwg), dim3(256), 0, st, d);
            else if (mode == 4) hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 128, AMODE, 4>), dim3((unsigned)nwg), dim3(256), 0, st, d);
            else hipLaunchKernelGGL((gemm_dma_kernel<T, 128, 128, AMODE, 3>), dim3((unsigned)nwg), dim3(256), 0, st, d);
            return lwdetr_check_launch();
        }
    }
    if (small) hipLaunchKernelGGL((gemm_kernel<T, 64, 64, AMODE>), dim3((unsigned)nwg), dim3(256), 0, st, d);
    else if (bn64) hipLaunchKernelGGL((gemm_kernel<T, 128, 64, AMODE>), dim3((unsigned)nwg), dim3(256), 0, st, d);
    else if constexpr (sizeof(T) == 2)
        hipLaunchKernelGGL((gemm_kernel<T, 128, 128, AMODE>), dim3((unsigned)nwg), dim3(256), 0, st, d);
    return lwdetr_check_launch();
}

template <typename T>
int dispatch_amode(const lwdetr_gemm_desc& d, hipStream_t st) {
    switch (d.a_mode) {
        case LWDETR_A_PLAIN: return launch<T, LWDETR_A_PLAIN>(d, st);
        case LWDETR_A_CONV3x3: return launch<T, LWDETR_A_CONV3x3>(d, st);
        case LWDETR_A_PATCH16: return launch<T, LWDETR_A_PATCH16>(d, st);
        default: return LWDETR_ERR_BAD_ARG;
    }
}

}  // namespace

extern "C" void lwdetr_gemm_tuning(int big_mode) { g_big_mode = big_mode; }
extern "C" int lwdetr_has_experiments(void) {
#ifdef LWDETR_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int lwdetr_gemm(const lwdetr_gemm_desc* desc, int dtype, void* hip_stream) {
    if (!desc) return LWDETR_ERR_BAD_ARG;
    const lwdetr_gemm_desc& d = *desc;
    if (d.M < 0 || d.N <= 0 || d.K <= 0 || d.K % BK != 0 || !d.A || !d.W) return LWDETR_ERR_BAD_ARG;
    if (d.M == 0) return LWDETR_OK;
    if (d.nseg < 1 || d.nseg > 3 || d.seg[0].n_begin != 0) return LWDETR_ERR_BAD_ARG;
    const int esz = dtype == DT_F32 ? 4 : 2;
    const int epc = 16 / esz;
    for (int s = 0; s < d.nseg; ++s) {
        const lwdetr_gemm_seg& g = d.seg[s];
        if (!g.out || g.n_end <= g.n_begin) return LWDETR_ERR_BAD_ARG;
        if (s + 1 < d.nseg && d.seg[s + 1].n_begin != g.n_end) return LWDETR_ERR_BAD_ARG;
        if ((g.mode == LWDETR_OUT_HEADS || g.mode == LWDETR_OUT_HEADS_T) &&
            (g.p0 <= 0 || g.p1 <= 0 || g.p2 <= 0 || g.p1 % 4 != 0 || g.p0 % 4 != 0)) return LWDETR_ERR_BAD_ARG;
        if (g.mode == LWDETR_OUT_DECONV2x2 && (g.p0 <= 0 || g.p0 % 4 != 0)) return LWDETR_ERR_BAD_ARG;
        if (g.mode < 0 || g.mode > LWDETR_OUT_DECONV2x2) return LWDETR_ERR_BAD_ARG;
        if ((g.ln_stats != nullptr) != (g.ln_colsum != nullptr)) return LWDETR_ERR_BAD_ARG;
        if (g.ln_stats && (d.a_mode != LWDETR_A_PLAIN || d.A2)) return LWDETR_ERR_BAD_ARG;      // row statistics of the plain A rows
        if ((g.ln_stats != nullptr) != (d.seg[0].ln_stats != nullptr)) return LWDETR_ERR_BAD_ARG;    // all segments of a launch or none
    }
    if (d.seg[d.nseg - 1].n_end < d.N) return LWDETR_ERR_BAD_ARG;
    if (d.splitk < 0 || d.splitk > 16 || (d.splitk >= 2 && (!d.splitk_ws || ((size_t)d.splitk_ws & 15) != 0))) return LWDETR_ERR_BAD_ARG;
    if (d.a_mode == LWDETR_A_PLAIN && (d.lda % epc != 0)) return LWDETR_ERR_BAD_ARG;
    if (d.a_mode == LWDETR_A_CONV3x3 &&
        (d.conv_cin % BK != 0 || d.K != 9 * d.conv_cin || (d.conv_stride != 1 && d.conv_stride != 2) ||
         d.lda % epc != 0 || d.a_col0 % epc != 0 || d.conv_hout <= 0 || d.conv_wout <= 0)) return LWDETR_ERR_BAD_ARG;
    if (d.a_mode == LWDETR_A_PATCH16 && (d.K != 768 || d.img_w % 16 != 0 || d.img_h % 16 != 0)) return LWDETR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F32: return dispatch_amode<float>(d, st);
        case DT_F16: return dispatch_amode<f16>(d, st);
        case DT_BF16: return dispatch_amode<bf16>(d, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}
