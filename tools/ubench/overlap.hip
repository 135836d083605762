// Does plain VALU / transcendental work issue in the shadow of an MFMA on gfx950? Hand-placed streams (inline asm, so
// hipcc cannot reorder): per group ONE v_mfma (4 accumulators round-robin: never dependent back to back) followed by
// NV v_mul_f32 + NT v_exp_f32 on registers the MFMA does not touch. Cycles per group by s_memtime, one block per CU,
// 1 / 2 / 3 waves per SIMD.    hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define REP2(x) x x
#define REP4(x) REP2(x) REP2(x)
#define REP8(x) REP4(x) REP4(x)
#define VMUL "v_mul_f32 %[v0], %[v0], %[one]\n\tv_mul_f32 %[v1], %[v1], %[one]\n\t"
#define VEXP "v_exp_f32 %[v2], %[v2]\n\t"

template <int KIND, int NV, int NT>      // KIND 0: 32x32x16, 1: 16x16x32, 2: no MFMA
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.25f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    float v0 = threadIdx.x * 0.01f, v1 = 1.f + threadIdx.x, v2 = -0.5f, one = 1.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define GROUP(C, D)                                                                                                      \
        if (KIND == 0) asm volatile("v_mfma_f32_32x32x16_f16 %[c], %[a], %[b], %[c]\n\t" : [c] "+v"(C) : [a] "v"(a), [b] "v"(b)); \
        if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_f16 %[c], %[a], %[b], %[c]\n\t" : [c] "+v"(D) : [a] "v"(a), [b] "v"(b)); \
        if (NV == 2) asm volatile(VMUL : [v0] "+v"(v0), [v1] "+v"(v1) : [one] "v"(one));                                      \
        if (NV == 4) asm volatile(REP2(VMUL) : [v0] "+v"(v0), [v1] "+v"(v1) : [one] "v"(one));                                \
        if (NV == 8) asm volatile(REP4(VMUL) : [v0] "+v"(v0), [v1] "+v"(v1) : [one] "v"(one));                                \
        if (NV == 16) asm volatile(REP8(VMUL) : [v0] "+v"(v0), [v1] "+v"(v1) : [one] "v"(one));                               \
        if (NT == 1) asm volatile(VEXP : [v2] "+v"(v2));                                                                     \
        if (NT == 2) asm volatile(REP2(VEXP) : [v2] "+v"(v2));                                                               \
        if (NT == 4) asm volatile(REP4(VEXP) : [v2] "+v"(v2));                                                               \
        if (NT == 8) asm volatile(REP8(VEXP) : [v2] "+v"(v2));
        GROUP(c0, d0) GROUP(c1, d1) GROUP(c2, d2) GROUP(c3, d3)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    for (int j = 0; j < 4; ++j) s += d0[j] + d1[j] + d2[j] + d3[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int KIND, int NV, int NT> void run(const char* what) {
    float* out; unsigned long long* cyc; unsigned long long h[16];
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 128);
    const int iters = 500;
    printf("%-44s", what);
    for (int threads : {256, 512, 768}) {
        hipMemset(cyc, 0, 128);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, NV, NT>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        double mx = 0; for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        printf("  %dw/SIMD: %6.1f", threads / 256, mx / (iters * 4.0));
    }
    printf("   cycles per group (slowest wave)\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0, 0>("mfma32x32x16");
    run<0, 2, 0>("mfma32x32x16 + 2 v_mul");
    run<0, 4, 0>("mfma32x32x16 + 4 v_mul");
    run<0, 8, 0>("mfma32x32x16 + 8 v_mul");
    run<0, 16, 0>("mfma32x32x16 + 16 v_mul");
    run<0, 0, 2>("mfma32x32x16 + 2 v_exp");
    run<0, 0, 4>("mfma32x32x16 + 4 v_exp");
    run<0, 0, 8>("mfma32x32x16 + 8 v_exp");
    run<0, 4, 4>("mfma32x32x16 + 4 v_mul + 4 v_exp");
    run<0, 8, 4>("mfma32x32x16 + 8 v_mul + 4 v_exp");
    run<1, 0, 0>("mfma16x16x32");
    run<1, 2, 0>("mfma16x16x32 + 2 v_mul");
    run<1, 4, 0>("mfma16x16x32 + 4 v_mul");
    run<1, 8, 0>("mfma16x16x32 + 8 v_mul");
    run<1, 0, 2>("mfma16x16x32 + 2 v_exp");
    run<1, 4, 2>("mfma16x16x32 + 4 v_mul + 2 v_exp");
    run<2, 8, 0>("8 v_mul alone");
    run<2, 16, 0>("16 v_mul alone");
    run<2, 0, 4>("4 v_exp alone");
    run<2, 0, 8>("8 v_exp alone");
    run<2, 8, 4>("8 v_mul + 4 v_exp alone");
    return 0;
}
