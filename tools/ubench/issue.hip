// Issue-rate microbenchmark for gfx950 (tuning tool): how many VALU / transcendental instructions hide under one
// v_mfma_f32_16x16x32_f16, within a wave and across the two waves of a SIMD.   hipcc --offload-arch=gfx950 -O3 issue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE: number of plain VALU (v_fma_f32, 4 independent chains) and transcendentals (v_exp_f32) after each MFMA
template <int NMFMA, int NVALU, int NTRANS, bool SPLIT>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    // SPLIT: waves 0-3 (first wave of each SIMD) issue only MFMAs, waves 4-7 only VALU/trans
    const bool do_m = !SPLIT || wave < 4, do_v = !SPLIT || wave >= 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (NMFMA && do_m) {
#pragma unroll
                for (int m = 0; m < NMFMA; ++m)
                    acc[(r + m) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[(r + m) & 7], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < NVALU; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
#pragma unroll
                for (int q = 0; q < NTRANS; ++q) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void kmf(float* out, unsigned long long* cyc, int iters) {
    f16x8 a8, b8; f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(i * 0.5f); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16 acc32[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) acc[r] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[r], 0, 0, 0);
            if (KIND == 1) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[r], 0, 0, 0);
            if (KIND == 2) acc32[r & 1] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc32[r & 1], 0, 0, 0);
            if (KIND == 3) acc32[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc32[r & 1], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc32[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void runk(const char* what) {
    float* out; unsigned long long* cyc; unsigned long long h[8] = {0};
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((kmf<KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-30s %.1f cycles per MFMA (1 wave/SIMD)\n", what, (double)h[0] / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

template <int NMFMA, int NVALU, int NTRANS, int NLDS>
__global__ __launch_bounds__(512) void k32(float* out, unsigned long long* cyc, int iters) {
    __shared__ float4 lds[1024];
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    lds[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f); lds[threadIdx.x + 512] = make_float4(1.f, 2.f, 3.f, 4.f);
    float4 ld[4] = {};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m)
                acc[(r + m) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[(r + m) & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NLDS; ++q) ld[q & 3] = lds[(threadIdx.x + 64 * (q + r)) & 1023];
#pragma unroll
            for (int q = 0; q < NVALU; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int q = 0; q < NTRANS; ++q) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += ld[i].x + ld[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int NMFMA, int NVALU, int NTRANS, int NLDS>
void run32(int threads, const char* what) {
    float* out; unsigned long long* cyc; unsigned long long h[8] = {0};
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64); hipMemset(cyc, 0, 64);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k32<NMFMA, NVALU, NTRANS, NLDS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-58s waves/SIMD %d  cycles per group: wave0 %.1f  wave4 %.1f\n", what, threads / 256, (double)h[0] / (iters * 8.0), (double)h[threads > 256 ? 4 : 0] / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

template <int NMFMA, int NVALU, int NTRANS, bool SPLIT>
void run(int threads, const char* what) {
    float* out; unsigned long long* cyc; unsigned long long h[8] = {0};
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64); hipMemset(cyc, 0, 64);
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NMFMA, NVALU, NTRANS, SPLIT>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (iters * 8.0), per4 = (double)h[threads > 256 ? 4 : 0] / (iters * 8.0);
    printf("%-58s waves/SIMD %d  cycles per group: wave0 %.1f  wave4 %.1f\n", what, threads / 256, per, per4);
    hipFree(out); hipFree(cyc);
}

int main() {
    runk<0>("16x16x16 f16 (legacy)"); runk<1>("16x16x32 f16"); runk<2>("32x32x8 f16 (legacy)"); runk<3>("32x32x16 f16");
    run<1, 0, 0, false>(256, "1 mfma16");
    run<1, 4, 0, false>(256, "1 mfma16 + 4 fma");
    run32<1, 0, 0, 0>(256, "1 mfma32x32x16");
    run32<1, 0, 0, 0>(512, "1 mfma32x32x16");
    run32<1, 2, 0, 0>(256, "1 mfma32 + 2 fma");
    run32<1, 4, 0, 0>(256, "1 mfma32 + 4 fma");
    run32<1, 6, 0, 0>(256, "1 mfma32 + 6 fma");
    run32<1, 8, 0, 0>(256, "1 mfma32 + 8 fma");
    run32<1, 12, 0, 0>(256, "1 mfma32 + 12 fma");
    run32<1, 4, 0, 0>(512, "1 mfma32 + 4 fma");
    run32<1, 8, 0, 0>(512, "1 mfma32 + 8 fma");
    run32<1, 4, 1, 0>(256, "1 mfma32 + 4 fma + 1 exp2");
    run32<1, 4, 2, 0>(256, "1 mfma32 + 4 fma + 2 exp2");
    run32<1, 6, 2, 0>(512, "1 mfma32 + 6 fma + 2 exp2");
    run32<1, 0, 0, 1>(256, "1 mfma32 + 1 ds_read_b128");
    run32<1, 0, 0, 2>(256, "1 mfma32 + 2 ds_read_b128");
    run32<1, 4, 0, 1>(256, "1 mfma32 + 4 fma + 1 ds_read_b128");
    run32<1, 4, 0, 1>(512, "1 mfma32 + 4 fma + 1 ds_read_b128");
    run32<1, 6, 2, 1>(512, "1 mfma32 + 6 fma + 2 exp2 + 1 ds_read");
    run32<0, 8, 0, 0>(256, "8 fma");
    run32<0, 0, 0, 4>(256, "4 ds_read_b128");
    return 0;
}
