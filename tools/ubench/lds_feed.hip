// LDS fragment-read throughput vs MFMA issue (tuning tool for the large-tile GEMM): per "chunk" a wave reads NR 1 KB operand
// fragments (ds_read_b128, the GEMM's swizzled 32-row pattern or lane-linear) and issues NM 32x32x16 MFMAs on 8 accumulators.
// Wall time by HIP events on a full-chip launch (256 workgroups of 512 or 256 threads).
//   hipcc --offload-arch=gfx950 -O3 lds_feed.hip -o lds_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NR, int NM, int PATTERN>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 128 * 1024 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    f16x8 fr[NR > 0 ? NR : 1];
    for (int r = 0; r < (NR > 0 ? NR : 1); ++r) for (int e = 0; e < 8; ++e) fr[r][e] = (_Float16)(lane + e);
    int base = (wave & 3) * 32 * 128;                        // a 32-row x 128-byte tile per wave (rows of KB = 64 halves)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int off = PATTERN == 0 ? base + r * 4096 + m * 128 + (((2 * c + h) ^ ((m >> 1) & 7)) << 4)
                                             : base + r * 4096 + c * 1024 + lane * 16;
                fr[r] = *(const f16x8*)(smem + (off & (128 * 1024 - 1)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NM; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[j % (NR > 0 ? NR : 1)], fr[(j + 1) % (NR > 0 ? NR : 1)], acc[j & 7], 0, 0, 0);
            if (NM == 0) { for (int r = 0; r < NR; ++r) asm volatile("" :: "v"(fr[r])); }
        }
        base = (base + 8192) & (64 * 1024 - 1);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 8-wave ping-pong: waves w and w + 4 share a SIMD; one group issues its fragment reads (memory cluster) while the other
// issues its MFMAs (compute cluster), swapping roles at every s_barrier.
template <int NR, int NM, bool PRIO>
__global__ __launch_bounds__(512) void kpp(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 128 * 1024 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    f16x8 fr[2][NR];
    for (int b = 0; b < 2; ++b) for (int r = 0; r < NR; ++r) for (int e = 0; e < 8; ++e) fr[b][r][e] = (_Float16)(lane + e);
    int base = (wave & 3) * 32 * 128;
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int off = base + r * 4096 + m * 128 + (((2 * c + h) ^ ((m >> 1) & 7)) << 4);
                fr[(c + 1) & 1][r] = *(const f16x8*)(smem + (off & (128 * 1024 - 1)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NM; ++j) acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[c & 1][j % NR], fr[c & 1][(j + 1) % NR], acc[j & 7], 0, 0, 0);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        base = (base + 8192) & (64 * 1024 - 1);
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NR, int NM, bool PRIO>
void runpp(const char* what, float* out) {
    (void)hipFuncSetAttribute((const void*)kpp<NR, NM, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, threads = 512;
    hipLaunchKernelGGL((kpp<NR, NM, PRIO>), dim3(256), dim3(threads), 128 * 1024, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kpp<NR, NM, PRIO>), dim3(256), dim3(threads), 128 * 1024, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double clk = ms * 1e-3 * 2.4e9, waves = threads / 64;
    printf("%-44s %d waves/CU: %7.1f us  LDS %6.1f B/clk/CU (at 2.4 GHz)  MFMA %7.1f TFLOP/s  %6.0f clk per 4-chunk step\n", what, (int)waves,
           ms * 1e3, waves * iters * 4.0 * NR * 1024 / clk, 256.0 * waves * iters * 4 * NM * 32768.0 / (ms * 1e-3) / 1e12, clk / iters);
}

template <int NR, int NM, int PATTERN>
void run(const char* what, float* out) {
    (void)hipFuncSetAttribute((const void*)k<NR, NM, PATTERN>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        const int iters = 4000;
        hipLaunchKernelGGL((k<NR, NM, PATTERN>), dim3(256), dim3(threads), 128 * 1024, 0, out, iters);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NR, NM, PATTERN>), dim3(256), dim3(threads), 128 * 1024, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double clk = ms * 1e-3 * 2.4e9, waves = threads / 64;
        printf("%-44s %d waves/CU: %7.1f us  LDS %6.1f B/clk/CU (at 2.4 GHz)  MFMA %7.1f TFLOP/s  %6.0f clk per 4-chunk step\n", what, (int)waves,
               ms * 1e3, waves * iters * 4.0 * NR * 1024 / clk, 256.0 * waves * iters * 4 * NM * 32768.0 / (ms * 1e-3) / 1e12, clk / iters);
    }
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<6, 0, 0>("reads only, 6 frags/chunk, GEMM swizzle", out);
    run<6, 0, 1>("reads only, 6 frags/chunk, lane-linear", out);
    run<0, 8, 0>("MFMA only, 8/chunk", out);
    run<6, 8, 0>("6 reads + 8 MFMA / chunk (256x256 tile, 8 waves)", out);
    run<6, 8, 1>("6 reads + 8 MFMA / chunk, lane-linear", out);
    run<8, 16, 0>("8 reads + 16 MFMA / chunk (128x128 wave tile)", out);
    run<4, 4, 0>("4 reads + 4 MFMA / chunk (64x64 wave tile)", out);
    run<3, 8, 0>("3 reads + 8 MFMA / chunk", out);
    runpp<6, 8, false>("PING-PONG 6 reads + 8 MFMA / chunk", out);
    runpp<6, 8, true>("PING-PONG 6 reads + 8 MFMA / chunk, setprio", out);
    runpp<4, 4, true>("PING-PONG 4 reads + 4 MFMA / chunk, setprio", out);
    return 0;
}
