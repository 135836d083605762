// Chip-wide sustained MFMA rate (tuning tool): every CU runs back-to-back v_mfma_f32_32x32x16_f16 on 8 independent accumulators
// per wave, 1 or 2 waves per SIMD; wall time by HIP events (-> TFLOP/s) and s_memtime inside (-> the clock the loop really ran at).
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); }
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc; unsigned long long h = 0;
    (void)hipMalloc(&out, 2048 * 512 * 4); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int threads : {256, 512}) for (int wgs : {256, 512, 1024}) for (int iters : {2000, 20000}) {
        hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double flop = (double)wgs * (threads / 64) * iters * 8.0 * 32768.0;
        printf("threads %4d WGs %5d iters %6d: %8.1f us  %7.1f TFLOP/s   loop %llu memtime ticks -> %.1f ticks per MFMA per wave, %.0f MHz if one tick = one clock\n",
               threads, wgs, iters, ms * 1e3, flop / (ms * 1e-3) / 1e12, h, (double)h / (iters * 8.0), (double)h / (ms * 1e3));
    }
    return 0;
}
