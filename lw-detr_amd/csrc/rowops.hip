// Row-wise LayerNorm for gfx950 (HBM-bound): 16 lanes per row, 4 rows per wave, 16-byte loads, statistics in f32
// with a two-pass (mean, then centred variance) formulation over register-resident values - the same arithmetic as
// nn.LayerNorm / the projector's channel LayerNorm (biased variance; reference models/backbone/projector.py:43-46).
#include "common.h"

namespace {

constexpr int LN_MAX_CHUNKS = 16;   // per lane: supports C <= 16 lanes * 16 chunks * EPC (2048 for 16-bit, 1024 for f32)

// Optional chained second LayerNorm (gamma2 != nullptr): out2 = LN2(out) computed on the values just rounded to T, i.e.
// bit-identical to a second launch reading `out` (decoder: norm3 followed by the shared decoder.norm, transformer.py:466-517,
// :397-400) - one pass over the row instead of two launches.
template <typename T, int NCH>   // NCH = 16-byte chunks held per lane (register resident row)
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, long ldx, const float* __restrict__ gamma,      // x / out: no __restrict__ -
                                                        const float* __restrict__ beta, T* out, long ldo,         // lwdetr_ffn_finish runs in place (out1 == x)
                                                        long M, int C, float eps, long rows_per_batch,
                                                        long out_batch_rows, long out_row_offset,
                                                        const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                        T* __restrict__ out2, long ldo2, float eps2,
                                                        const float* __restrict__ part, int nsplit,
                                                        const float* __restrict__ pbias) {
    constexpr int EPC = 16 / (int)sizeof(T);
    typedef T VC __attribute__((ext_vector_type(EPC)));
    const int lane16 = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= M) return;
    const int nchunks = C / EPC;
    const T* xr = x + row * ldx;
    float v[NCH][EPC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane16 + 16 * i;
        if (c < nchunks) {
            const VC t = *(const VC*)(xr + c * EPC);
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[i][e] = to_f32<T>(t[e]);
            if (part) {     // FFN finish: row = T(x + bias + sum of the f32 split-hidden partial products), then the LayerNorms
                float a[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) a[e] = pbias[c * EPC + e];
#pragma unroll 4
                for (int sidx = 0; sidx < nsplit; ++sidx) {
                    const float* pr = part + ((long)sidx * M + row) * C + c * EPC;
#pragma unroll
                    for (int e4 = 0; e4 < EPC; e4 += 4) {
                        const f32x4 q = *(const f32x4*)(pr + e4);
                        a[e4] += q[0]; a[e4 + 1] += q[1]; a[e4 + 2] += q[2]; a[e4 + 3] += q[3];
                    }
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[i][e] = to_f32<T>(from_f32<T>(v[i][e] + a[e]));
            }
#pragma unroll
            for (int e = 0; e < EPC; ++e) sum += v[i][e];
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane16 + 16 * i;
        if (c < nchunks) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float dlt = v[i][e] - mean; var += dlt * dlt; }
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.f / sqrtf(var / (float)C + eps);
    long orow = row;
    if (rows_per_batch > 0) {
        const long b = row / rows_per_batch;
        orow = b * out_batch_rows + out_row_offset + (row - b * rows_per_batch);
    }
    T* outr = out + orow * ldo;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane16 + 16 * i;
        if (c < nchunks) {
            VC t;
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                t[e] = from_f32<T>((v[i][e] - mean) * rstd * gamma[c * EPC + e] + beta[c * EPC + e]);
            *(VC*)(outr + c * EPC) = t;
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[i][e] = to_f32<T>(t[e]);
        }
    }
    if (!gamma2) return;
    float sum2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane16 + 16 * i;
        if (c < nchunks) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) sum2 += v[i][e];
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) sum2 += __shfl_xor(sum2, o);
    const float mean2 = sum2 / (float)C;
    float var2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane16 + 16 * i;
        if (c < nchunks) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float dlt = v[i][e] - mean2; var2 += dlt * dlt; }
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) var2 += __shfl_xor(var2, o);
    const float rstd2 = 1.f / sqrtf(var2 / (float)C + eps2);
    T* outr2 = out2 + row * ldo2;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane16 + 16 * i;
        if (c < nchunks) {
            VC t;
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                t[e] = from_f32<T>((v[i][e] - mean2) * rstd2 * gamma2[c * EPC + e] + beta2[c * EPC + e]);
            *(VC*)(outr2 + c * EPC) = t;
        }
    }
}

template <typename T>
int launch_ln(const void* x, long ldx, const float* gamma, const float* beta, void* out, long ldo, long M, int C,
              float eps, long rpb, long obr, long oro, hipStream_t st, const float* gamma2 = nullptr,
              const float* beta2 = nullptr, void* out2 = nullptr, long ldo2 = 0, float eps2 = 0.f,
              const float* part = nullptr, int nsplit = 0, const float* pbias = nullptr) {
    constexpr int EPC = 16 / (int)sizeof(T);
    if (C % EPC != 0 || C / EPC > 16 * LN_MAX_CHUNKS || ldx % EPC != 0 || ldo % EPC != 0 || ldo2 % EPC != 0) return LWDETR_ERR_UNSUPPORTED;
    const long blocks = (M + 15) / 16;
    ProfScope ps(KID_LAYERNORM, 0.0, (gamma2 ? 3.0 : 2.0) * M * C * sizeof(T) + 4.0 * nsplit * M * C, st);
    const int nch = (C / EPC + 15) / 16;
#define LN_LAUNCH(N) hipLaunchKernelGGL((layernorm_kernel<T, N>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, \
                                        ldx, gamma, beta, (T*)out, ldo, M, C, eps, rpb, obr, oro, gamma2, beta2, (T*)out2, ldo2, eps2, \
                                        part, nsplit, pbias)
    if (nch <= 2) LN_LAUNCH(2);
    else if (nch <= 3) LN_LAUNCH(3);
    else if (nch <= 4) LN_LAUNCH(4);
    else if (nch <= 6) LN_LAUNCH(6);
    else if (nch <= 8) LN_LAUNCH(8);
    else if (nch <= 12) LN_LAUNCH(12);
    else LN_LAUNCH(16);
#undef LN_LAUNCH
    return lwdetr_check_launch();
}

// Row statistics only (round 5): the first two passes of layernorm_kernel - (mean, rstd) of every row for the LayerNorm-folded GEMM
// epilogue (gemm.hip: ln_stats) - without the normalised copy of x that lwdetr_layernorm writes and the next GEMM reads back.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, long ldx, long M, int C, float eps, float* __restrict__ stats) {
    constexpr int EPC = 16 / (int)sizeof(T);
    typedef T VC __attribute__((ext_vector_type(EPC)));
    const int lane16 = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= M) return;
    const int nchunks = C / EPC;
    const T* xr = x + row * ldx;
    float v[NCH][EPC];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane16 + 16 * i;
        if (c < nchunks) {
            const VC t = *(const VC*)(xr + c * EPC);
#pragma unroll
            for (int e = 0; e < EPC; ++e) { v[i][e] = to_f32<T>(t[e]); sum += v[i][e]; }
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane16 + 16 * i;
        if (c < nchunks) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) { const float dlt = v[i][e] - mean; var += dlt * dlt; }
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) var += __shfl_xor(var, o);
    if (lane16 == 0) { stats[row] = mean; stats[M + row] = 1.f / sqrtf(var / (float)C + eps); }
}
template <typename T>
int launch_row_stats(const void* x, long ldx, long M, int C, float eps, float* stats, hipStream_t st) {
    constexpr int EPC = 16 / (int)sizeof(T);
    if (C % EPC != 0 || C / EPC > 16 * LN_MAX_CHUNKS || ldx % EPC != 0) return LWDETR_ERR_UNSUPPORTED;
    const long blocks = (M + 15) / 16;
    ProfScope ps(KID_LAYERNORM, 0.0, 1.0 * M * C * sizeof(T), st);
    const int nch = (C / EPC + 15) / 16;
#define RS_LAUNCH(N) hipLaunchKernelGGL((row_stats_kernel<T, N>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, ldx, M, C, eps, stats)
    if (nch <= 2) RS_LAUNCH(2);
    else if (nch <= 3) RS_LAUNCH(3);
    else if (nch <= 4) RS_LAUNCH(4);
    else if (nch <= 6) RS_LAUNCH(6);
    else if (nch <= 8) RS_LAUNCH(8);
    else if (nch <= 12) RS_LAUNCH(12);
    else RS_LAUNCH(16);
#undef RS_LAUNCH
    return lwdetr_check_launch();
}

}  // namespace

extern "C" int lwdetr_row_stats(const void* x, long ldx, long M, int C, float eps, float* stats, int dtype, void* hip_stream) {
    if (!x || !stats || M < 0 || C <= 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F32: return launch_row_stats<float>(x, ldx, M, C, eps, stats, st);
        case DT_F16: return launch_row_stats<f16>(x, ldx, M, C, eps, stats, st);
        case DT_BF16: return launch_row_stats<bf16>(x, ldx, M, C, eps, stats, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

extern "C" int lwdetr_layernorm(const void* x, long ldx, const float* gamma, const float* beta, void* out, long ldo,
                                long M, int C, float eps, long rows_per_batch, long out_batch_rows, long out_row_offset,
                                int dtype, void* hip_stream) {
    if (!x || !gamma || !beta || !out || M < 0 || C <= 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F32: return launch_ln<float>(x, ldx, gamma, beta, out, ldo, M, C, eps, rows_per_batch, out_batch_rows, out_row_offset, st);
        case DT_F16: return launch_ln<f16>(x, ldx, gamma, beta, out, ldo, M, C, eps, rows_per_batch, out_batch_rows, out_row_offset, st);
        case DT_BF16: return launch_ln<bf16>(x, ldx, gamma, beta, out, ldo, M, C, eps, rows_per_batch, out_batch_rows, out_row_offset, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

extern "C" int lwdetr_layernorm_chain(const void* x, long ldx, const float* gamma1, const float* beta1, float eps1, void* out1,
                                      long ldo1, const float* gamma2, const float* beta2, float eps2, void* out2, long ldo2,
                                      long M, int C, int dtype, void* hip_stream) {
    if (!x || !gamma1 || !beta1 || !out1 || !gamma2 || !beta2 || !out2 || M < 0 || C <= 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F32: return launch_ln<float>(x, ldx, gamma1, beta1, out1, ldo1, M, C, eps1, 0, 0, 0, st, gamma2, beta2, out2, ldo2, eps2);
        case DT_F16: return launch_ln<f16>(x, ldx, gamma1, beta1, out1, ldo1, M, C, eps1, 0, 0, 0, st, gamma2, beta2, out2, ldo2, eps2);
        case DT_BF16: return launch_ln<bf16>(x, ldx, gamma1, beta1, out1, ldo1, M, C, eps1, 0, 0, 0, st, gamma2, beta2, out2, ldo2, eps2);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

// decoder FFN, second half: x <- T(x + b2 + sum_s partial[s]); out1 = norm3(x); out2 = decoder.norm(out1) (16-bit only:
// the producer, lwdetr_ffn_partial, runs on the 16-bit MFMA block kernel)
extern "C" int lwdetr_ffn_finish(const void* x, long ldx, const float* partial, int splits, const float* b2, const float* gamma1,
                                 const float* beta1, float eps1, void* out1, long ldo1, const float* gamma2, const float* beta2,
                                 float eps2, void* out2, long ldo2, long M, int C, int dtype, void* hip_stream) {
    if (!x || !partial || splits <= 0 || !b2 || !gamma1 || !beta1 || !out1 || M < 0 || C <= 0 || C % 8 != 0) return LWDETR_ERR_BAD_ARG;
    if (gamma2 && (!beta2 || !out2)) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F16: return launch_ln<f16>(x, ldx, gamma1, beta1, out1, ldo1, M, C, eps1, 0, 0, 0, st, gamma2, beta2, out2, ldo2, eps2, partial, splits, b2);
        case DT_BF16: return launch_ln<bf16>(x, ldx, gamma1, beta1, out1, ldo1, M, C, eps1, 0, 0, 0, st, gamma2, beta2, out2, ldo2, eps2, partial, splits, b2);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Small fused glue kernels of the two-stage selection / decoder set-up (each replaces a dozen tiny tensor ops):
//   select_gather   : rows picked by the two-stage top-k -> encoder features, their class logits, their anchor proposals
//                     (reference models/transformer.py:248-255, models/lwdetr.py:168-170)
//   decoder_inputs  : box re-parameterisation of the selected proposals and of the learned reference points
//                     (transformer.py:236-240, :266-274), sine embedding of the references (transformer.py:42-68, order
//                     y,x,w,h, applied to ref * valid_ratio of level 0, :352-355) and the query broadcast (:266)
//   box_reparam     : final boxes of all decoder layers (models/lwdetr.py:150-155)
namespace {

template <typename T>
__global__ __launch_bounds__(256) void select_gather_kernel(const T* __restrict__ om, const T* __restrict__ enc_cls, long ldc,
                                                            const float* __restrict__ props, const int64_t* __restrict__ idx,
                                                            T* __restrict__ om_sel, T* __restrict__ logits_out,
                                                            float* __restrict__ props_sel, int B, int S, int d, int nq, int ncls) {
    const long row = blockIdx.x;                       // b * nq + q
    const int b = (int)(row / nq);
    const long src = (long)b * S + idx[row];
    for (int c = threadIdx.x; c < d; c += blockDim.x) om_sel[row * d + c] = om[src * d + c];
    for (int c = threadIdx.x; c < ncls; c += blockDim.x) logits_out[row * ncls + c] = enc_cls[src * ldc + c];
    if (threadIdx.x < 4) props_sel[row * 4 + threadIdx.x] = props[src * 4 + threadIdx.x];
}

template <typename T>
__global__ __launch_bounds__(256) void decoder_inputs_kernel(const T* __restrict__ enc_delta, const float* __restrict__ props_sel,
                                                             const float* __restrict__ refpoint, const float* __restrict__ vr,
                                                             int L, const T* __restrict__ query_feat,
                                                             const float* __restrict__ dim_t, T* __restrict__ enc_boxes,
                                                             float* __restrict__ ref_out, T* __restrict__ sine,
                                                             T* __restrict__ xdec, int B, int nq, int d) {
    const long row = blockIdx.x;
    const int b = (int)(row / nq), q = (int)(row - (long)b * nq);
    float dl[4], pr[4], ts[4], rp[4], rf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dl[i] = to_f32<T>(enc_delta[row * 4 + i]); pr[i] = props_sel[row * 4 + i]; rp[i] = refpoint[q * 4 + i]; }
    ts[0] = dl[0] * pr[2] + pr[0]; ts[1] = dl[1] * pr[3] + pr[1]; ts[2] = expf(dl[2]) * pr[2]; ts[3] = expf(dl[3]) * pr[3];
    rf[0] = rp[0] * ts[2] + ts[0]; rf[1] = rp[1] * ts[3] + ts[1]; rf[2] = expf(rp[2]) * ts[2]; rf[3] = expf(rp[3]) * ts[3];
    if (threadIdx.x < 4) { enc_boxes[row * 4 + threadIdx.x] = from_f32<T>(ts[threadIdx.x]); ref_out[row * 4 + threadIdx.x] = rf[threadIdx.x]; }
    const float vx = vr[(long)b * L * 2], vy = vr[(long)b * L * 2 + 1];
    const float pos[4] = {rf[1] * vy, rf[0] * vx, rf[2] * vx, rf[3] * vy};      // order (y, x, w, h)
    const int half = d / 2;
    for (int c = threadIdx.x; c < 2 * d; c += blockDim.x) {
        const int k = c / half, i = c - k * half;
        const float e = pos[k] * 6.283185307179586f / dim_t[i];
        sine[row * 2 * d + c] = from_f32<T>((i & 1) ? __cosf(e) : __sinf(e));   // |e| <= a few 2 pi: v_sin/v_cos, ~1e-6 abs
    }
    for (int c = threadIdx.x; c < d; c += blockDim.x) xdec[row * d + c] = query_feat[(long)q * d + c];
}

template <typename T>
__global__ __launch_bounds__(256) void box_reparam_kernel(const T* __restrict__ delta, const float* __restrict__ ref, long ref_rows,
                                                          T* __restrict__ out, long R) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float* rf = ref + (r % ref_rows) * 4;
    const float d0 = to_f32<T>(delta[r * 4]), d1 = to_f32<T>(delta[r * 4 + 1]), d2 = to_f32<T>(delta[r * 4 + 2]), d3 = to_f32<T>(delta[r * 4 + 3]);
    out[r * 4] = from_f32<T>(d0 * rf[2] + rf[0]);
    out[r * 4 + 1] = from_f32<T>(d1 * rf[3] + rf[1]);
    out[r * 4 + 2] = from_f32<T>(expf(d2) * rf[2]);
    out[r * 4 + 3] = from_f32<T>(expf(d3) * rf[3]);
}

// final boxes of all decoder layers (box_reparam) AND the contiguous copy of their class logits out of the padded GEMM
// output, in one launch: element i of the R x ncls logits block is copied by thread i, threads i < R also do row i's box.
// Input row r = layer * ref_rows + k lands in output row layer * out_layer_rows + k: with out_layer_rows > ref_rows the call
// fills its images' rows of a (layers, B_total, nq, .) tensor that other launch chains fill the rest of.
template <typename T>
__global__ __launch_bounds__(256) void finalize_outputs_kernel(const T* __restrict__ delta, const float* __restrict__ ref, long ref_rows,
                                                               T* __restrict__ coord, long R, const T* __restrict__ logits_pad,
                                                               long ldc, int ncls, T* __restrict__ logits_out, long layer_gap) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * ncls) {
        const long r = i / ncls;
        const long orow = r + (r / ref_rows) * layer_gap;
        logits_out[orow * ncls + (i - r * ncls)] = logits_pad[r * ldc + (i - r * ncls)];
    }
    if (i < R) {
        const long k = i % ref_rows;
        const float* rf = ref + k * 4;
        T* o = coord + (i + (i / ref_rows) * layer_gap) * 4;
        const float d0 = to_f32<T>(delta[i * 4]), d1 = to_f32<T>(delta[i * 4 + 1]), d2 = to_f32<T>(delta[i * 4 + 2]), d3 = to_f32<T>(delta[i * 4 + 3]);
        o[0] = from_f32<T>(d0 * rf[2] + rf[0]);
        o[1] = from_f32<T>(d1 * rf[3] + rf[1]);
        o[2] = from_f32<T>(expf(d2) * rf[2]);
        o[3] = from_f32<T>(expf(d3) * rf[3]);
    }
}

}  // namespace

#define LWDETR_DISPATCH_T(dtype, CALL)                 \
    switch (dtype) {                                   \
        case DT_F32: { typedef float TT; CALL; break; } \
        case DT_F16: { typedef f16 TT; CALL; break; }   \
        case DT_BF16: { typedef bf16 TT; CALL; break; } \
        default: return LWDETR_ERR_UNSUPPORTED;        \
    }

extern "C" int lwdetr_select_gather(const void* om, const void* enc_cls, long ldc, const float* props, const int64_t* idx,
                                    void* om_sel, void* logits_out, float* props_sel, int B, int S, int d, int nq, int ncls,
                                    int dtype, void* hip_stream) {
    if (!om || !enc_cls || !props || !idx || !om_sel || !logits_out || !props_sel || B <= 0 || nq <= 0) return LWDETR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    ProfScope ps(KID_ELTWISE, 0.0, 2.0 * B * nq * (d + ncls) * 2, st);
    LWDETR_DISPATCH_T(dtype, hipLaunchKernelGGL((select_gather_kernel<TT>), dim3(B * nq), dim3(256), 0, st, (const TT*)om,
                                                (const TT*)enc_cls, ldc, props, idx, (TT*)om_sel, (TT*)logits_out, props_sel, B,
                                                S, d, nq, ncls));
    return lwdetr_check_launch();
}

extern "C" int lwdetr_decoder_inputs(const void* enc_delta, const float* props_sel, const float* refpoint, const float* valid_ratios,
                                     int L, const void* query_feat, const float* dim_t, void* enc_boxes_out, float* ref_out,
                                     void* sine_out, void* xdec_out, int B, int nq, int d, int dtype, void* hip_stream) {
    if (!enc_delta || !props_sel || !refpoint || !valid_ratios || !query_feat || !dim_t || !enc_boxes_out || !ref_out ||
        !sine_out || !xdec_out || B <= 0 || nq <= 0 || d % 2) return LWDETR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    ProfScope ps(KID_ELTWISE, 0.0, 3.0 * B * nq * d * 2, st);
    LWDETR_DISPATCH_T(dtype, hipLaunchKernelGGL((decoder_inputs_kernel<TT>), dim3(B * nq), dim3(256), 0, st, (const TT*)enc_delta,
                                                props_sel, refpoint, valid_ratios, L, (const TT*)query_feat, dim_t,
                                                (TT*)enc_boxes_out, ref_out, (TT*)sine_out, (TT*)xdec_out, B, nq, d));
    return lwdetr_check_launch();
}

extern "C" int lwdetr_box_reparam(const void* delta, const float* ref, long ref_rows, void* out, long R, int dtype, void* hip_stream) {
    if (!delta || !ref || !out || R < 0 || ref_rows <= 0) return LWDETR_ERR_BAD_ARG;
    if (R == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    ProfScope ps(KID_ELTWISE, 0.0, 0.0, st);
    LWDETR_DISPATCH_T(dtype, hipLaunchKernelGGL((box_reparam_kernel<TT>), dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st,
                                                (const TT*)delta, ref, ref_rows, (TT*)out, R));
    return lwdetr_check_launch();
}

extern "C" int lwdetr_finalize_outputs(const void* delta, const float* ref, long ref_rows, void* coord_out, long R,
                                       const void* logits_pad, long ldc, int ncls, void* logits_out, long out_layer_rows, int dtype,
                                       void* hip_stream) {
    if (!delta || !ref || !coord_out || !logits_pad || !logits_out || R < 0 || ref_rows <= 0 || ncls <= 0 || ldc < ncls) return LWDETR_ERR_BAD_ARG;
    if (out_layer_rows == 0) out_layer_rows = ref_rows;
    if (out_layer_rows < ref_rows) return LWDETR_ERR_BAD_ARG;
    if (R == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    ProfScope ps(KID_ELTWISE, 0.0, 0.0, st);
    const long n = R * ncls;
    LWDETR_DISPATCH_T(dtype, hipLaunchKernelGGL((finalize_outputs_kernel<TT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                                                (const TT*)delta, ref, ref_rows, (TT*)coord_out, R, (const TT*)logits_pad, ldc, ncls,
                                                (TT*)logits_out, out_layer_rows - ref_rows));
    return lwdetr_check_launch();
}
