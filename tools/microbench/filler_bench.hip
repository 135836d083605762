// Tuning tool (not part of the product): what ONE kind of vector instruction costs when it sits between 32x32x16 MFMAs of a wave that has its
// SIMD to itself (the block kernel's situation: one wave per SIMD, GELU between the matrix instructions of the hidden loop).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_timing/filler_bench tools/microbench/filler_bench.hip ; run on the GPU: prints cycles per MFMA gap
//   for N = 0 .. 8 fillers of each kind per gap (s_memtime around 64 x 8 MFMAs on four accumulators, wave 0 of workgroup 0; 4 waves per
//   workgroup = one per SIMD, one workgroup per CU on every CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 4096
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FILL(K, x, y)                                                                                                      \
    do {                                                                                                                   \
        if constexpr (K == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));                                 \
        else if constexpr (K == 2) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x) : "v"(y));                             \
        else if constexpr (K == 3) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(x) : "v"(y));                         \
        else if constexpr (K == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x));                                             \
        else if constexpr (K == 5) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(x)); \
        else if constexpr (K == 6) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(y));                         \
        else if constexpr (K == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x2) : "v"(y2));                           \
        else if constexpr (K == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));                                             \
        else if constexpr (K == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));                                \
        else if constexpr (K == 10) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x) : "v"(y));                            \
        else if constexpr (K == 11) asm volatile("v_exp_f16 %0, %0" : "+v"(x));                                            \
        else if constexpr (K == 12) asm volatile("v_rcp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(x)); \
        else if constexpr (K == 13) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(y2));                      \
        else if constexpr (K == 14) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(agv));                     \
        else if constexpr (K == 15) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));                       \
        else if constexpr (K == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));                                   \
        else if constexpr (K == 17) asm volatile("s_nop 0");                                                               \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int K, int N>
__global__ __launch_bounds__(256, 1) void fb(unsigned long long* out, float* sink) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x16 acc[4] = {};
    unsigned r[8];
    f32x2 r2[8];
    for (int i = 0; i < 8; ++i) { r[i] = 0x3c003800u + lane + i; r2[i] = f32x2{0.5f + lane, 0.25f + i}; }
    unsigned y = 0x3c003c00u;
    f32x2 y2 = {1.0f, 1.0f};
    asm volatile("" : "+v"(y), "+v"(y2));
    float agv;
    asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(agv));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < N; ++f) {
                unsigned& x = r[(m * N + f) & 7];
                f32x2& x2 = r2[(m * N + f) & 7];
                (void)x; (void)x2;
                FILL(K, x, y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; if (blockIdx.x == 0) out[gridDim.x] = w1 - w0; }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 8; ++i) s += (float)r[i] + r2[i][0] + r2[i][1];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

static const char* NAMES[] = {"none", "v_fma_f32", "v_pk_mul_f16", "v_pk_fma_f16", "v_exp_f32", "v_exp_f16_sdwa", "v_cvt_pk_f16_f32", "v_pk_mul_f32",
                              "v_rcp_f32", "v_mul_f32", "v_pk_add_f16", "v_exp_f16", "v_rcp_f16_sdwa", "v_pk_fma_f32", "v_accvgpr_read", "v_permlane32_swap", "v_mov_b32", "s_nop"};

static double g_mhz = 0;
template <int K, int N>
double run(unsigned long long* d_out, float* d_sink, int grid) {
    std::vector<unsigned long long> h(grid + 1);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((fb<K, N>), dim3(grid), dim3(256), 0, 0, d_out, d_sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, (grid + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        g_mhz = (double)h[0] / ((double)h[grid] * 0.01);
        double v = (double)h[0] / (ITERS * 8.0);
        if (v < best) best = v;
    }
    return best;
}

template <int K>
void row(unsigned long long* d_out, float* d_sink, int grid) {
    printf("%-18s", NAMES[K]);
    printf(" %7.1f", run<K, 1>(d_out, d_sink, grid));
    printf(" %7.1f", run<K, 2>(d_out, d_sink, grid));
    printf(" %7.1f", run<K, 3>(d_out, d_sink, grid));
    printf(" %7.1f", run<K, 4>(d_out, d_sink, grid));
    printf(" %7.1f", run<K, 5>(d_out, d_sink, grid));
    printf(" %7.1f", run<K, 6>(d_out, d_sink, grid));
    printf(" %7.1f", run<K, 8>(d_out, d_sink, grid));
    printf("   %.0f MHz\n", g_mhz);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, (grid + 1) * sizeof(unsigned long long)); hipMalloc(&d_sink, 1024);
    printf("s_memtime ticks (100 MHz) would be useless here: s_memtime counts shader clocks on gfx950 (guide). cycles per MFMA gap, grid %d x 256 threads\n", grid);
    printf("%-18s %7s %7s %7s %7s %7s %7s %7s   (fillers per gap)\n", "filler", "1", "2", "3", "4", "5", "6", "8");
    printf("%-18s %7.1f   (shader clock during the loop: %.0f MHz = s_memtime / s_memrealtime)\n", "none", run<0, 1>(d_out, d_sink, grid), g_mhz);
    row<1>(d_out, d_sink, grid); row<9>(d_out, d_sink, grid); row<16>(d_out, d_sink, grid); row<17>(d_out, d_sink, grid);
    row<2>(d_out, d_sink, grid); row<3>(d_out, d_sink, grid); row<10>(d_out, d_sink, grid);
    row<4>(d_out, d_sink, grid); row<8>(d_out, d_sink, grid); row<11>(d_out, d_sink, grid); row<5>(d_out, d_sink, grid); row<12>(d_out, d_sink, grid);
    row<6>(d_out, d_sink, grid); row<7>(d_out, d_sink, grid); row<13>(d_out, d_sink, grid); row<14>(d_out, d_sink, grid); row<15>(d_out, d_sink, grid);
    return 0;
}
