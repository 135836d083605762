// Row-wise LayerNorm for gfx950 (HBM-bound): 16 lanes per row, 4 rows per wave, 16-byte loads, statistics in f32
// with a two-pass (mean, then centred variance) formulation over register-resident values - the same arithmetic as
// nn.LayerNorm / the projector's channel LayerNorm (biased variance; reference models/backbone/projector.py:43-46).
#include "common.h"

namespace {

constexpr int LN_MAX_CHUNKS = 16;   // per lane: supports C <= 16 lanes * 16 chunks * EPC (2048 for 16-bit, 1024 for f32)

template <typename T, int NCH>   // NCH = 16-byte chunks held per lane (register resident row)
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, long ldo,
                                                        long M, int C, float eps, long rows_per_batch,
                                                        long out_batch_rows, long out_row_offset) {
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
        }
    }
}

template <typename T>
int launch_ln(const void* x, long ldx, const float* gamma, const float* beta, void* out, long ldo, long M, int C,
              float eps, long rpb, long obr, long oro, hipStream_t st) {
    constexpr int EPC = 16 / (int)sizeof(T);
    if (C % EPC != 0 || C / EPC > 16 * LN_MAX_CHUNKS || ldx % EPC != 0 || ldo % EPC != 0) return LWDETR_ERR_UNSUPPORTED;
    const long blocks = (M + 15) / 16;
    ProfScope ps(KID_LAYERNORM, 0.0, 2.0 * M * C * sizeof(T), st);
    const int nch = (C / EPC + 15) / 16;
#define LN_LAUNCH(N) hipLaunchKernelGGL((layernorm_kernel<T, N>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, \
                                        ldx, gamma, beta, (T*)out, ldo, M, C, eps, rpb, obr, oro)
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

}  // namespace

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
