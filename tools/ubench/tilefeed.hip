// Operand-feed microbenchmark (tuning tool for the large-tile GEMM): the DMA pattern of a 256 x 256 tile and nothing else.
// Every workgroup (512 threads, one per CU) walks K in stages of 64 halves: per stage 256 "A" rows (its own, lda apart) and
// 256 "W" rows (shared by all workgroups), 128 bytes each, as global_load_lds_dwordx4 pieces of 8 rows x 128 B, DEPTH stages
// in flight (s_waitcnt vmcnt + optional s_barrier per stage). Wall time by HIP events -> bytes per clock and CU at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tilefeed.hip -o tilefeed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SRC: 0 = A rows of this workgroup's own tile (streams from HBM), 1 = every workgroup reads the tile of workgroup 0 (L2)
template <int DEPTH, bool BARRIER, int SRC, bool A_ONLY>
__global__ __launch_bounds__(512) void k(const char* __restrict__ A, long lda, const char* __restrict__ W, long ldw, int nk, int tiles_per_wg,
                                         float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    constexpr int PIECES = A_ONLY ? 4 : 8;                       // per wave and stage: 4 A pieces (+ 4 W pieces) of 8 rows
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = SRC == 1 ? 0 : (long)blockIdx.x * tiles_per_wg + t;
        const char* src[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = 8 * (wave + 8 * (p & 3)) + (lane >> 3);           // 0 .. 255
            src[p] = p < 4 ? A + (tile * 256 + row) * lda + (lane & 7) * 16 : W + (long)row * ldw + (lane & 7) * 16;
        }
        auto issue = [&](int kt) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const unsigned dst = lds0 + (unsigned)((kt % (DEPTH + 1)) * 65536 / (A_ONLY ? 2 : 1) + (p * 8 + wave) * 1024);
                const unsigned m0v = __builtin_amdgcn_readfirstlane(dst);
                const char* s = src[p] + (long)(kt < nk ? kt : 0) * 128;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(s) : "memory");
            }
        };
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) issue(s);
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt<(DEPTH - 1) * PIECES>();
            if (BARRIER) __builtin_amdgcn_s_barrier();
            issue(kt + DEPTH);
        }
        wait_vmcnt<0>();
        __syncthreads();
    }
    out[blockIdx.x * 512 + tid] = ((float*)smem)[tid];
}

template <int DEPTH, bool BARRIER, int SRC, bool A_ONLY>
void run(const char* what, const char* A, long lda, const char* W, long ldw, int nk, int tiles, float* out) {
    const int lds = 160 * 1024;
    (void)hipFuncSetAttribute((const void*)k<DEPTH, BARRIER, SRC, A_ONLY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<DEPTH, BARRIER, SRC, A_ONLY>), dim3(256), dim3(512), lds, 0, A, lda, W, ldw, nk, tiles, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<DEPTH, BARRIER, SRC, A_ONLY>), dim3(256), dim3(512), lds, 0, A, lda, W, ldw, nk, tiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double clk = ms * 1e-3 * 2.4e9, stages = (double)tiles * nk, bytes = stages * (A_ONLY ? 32768.0 : 65536.0);
    printf("%-58s depth %d%s: %7.1f us  %5.1f B/clk/CU  %6.0f clk per stage  (%.2f TB/s chip)\n", what, DEPTH, BARRIER ? " +barrier" : "         ",
           ms * 1e3, bytes / clk, clk / stages, bytes * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    const long K = 3072, M = 256L * 256 * 3;                     // 3 tiles per workgroup: A = 196608 x 3072 halves = 1.2 GB
    char *A, *W; float* out;
    (void)hipMalloc(&A, M * (K * 2 + 128)); (void)hipMalloc(&W, 256 * K * 2 * 16); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemset(A, 0, M * (K * 2 + 128)); (void)hipMemset(W, 0, 256 * K * 2 * 16);
    const int nk = (int)(K / 64);
#define ALL(SRC, AONLY, WHAT)                                              \
    run<1, true, SRC, AONLY>(WHAT, A, K * 2, W, K * 2, nk, 3, out);        \
    run<1, false, SRC, AONLY>(WHAT, A, K * 2, W, K * 2, nk, 3, out);       \
    run<2, true, SRC, AONLY>(WHAT, A, K * 2, W, K * 2, nk, 3, out);        \
    run<2, false, SRC, AONLY>(WHAT, A, K * 2, W, K * 2, nk, 3, out);
    ALL(0, false, "A own rows (HBM stream) + shared W, lda 6144 B")
    ALL(1, false, "A rows of tile 0 for everybody (L2) + shared W")
    ALL(0, true, "A own rows only (HBM stream)")
    run<4, false, 0, true>("A own rows only (HBM stream)", A, K * 2, W, K * 2, nk, 3, out);
    run<4, true, 0, true>("A own rows only (HBM stream)", A, K * 2, W, K * 2, nk, 3, out);
    // K = 768 geometry (row stride 1536 B): 12 stages per tile, 12 tiles per workgroup
    run<1, true, 0, false>("lda 1536 B: A own rows + shared W", A, 768 * 2, W, 768 * 2, 12, 12, out);
    run<2, true, 0, false>("lda 1536 B: A own rows + shared W", A, 768 * 2, W, 768 * 2, 12, 12, out);
    // padded row stride (6144 + 128 B): breaks a power-of-two channel pattern if there is one
    run<1, true, 0, false>("lda 6272 B: A own rows + shared W", A, K * 2 + 128, W, K * 2 + 128, nk, 3, out);
    run<2, true, 0, false>("lda 6272 B: A own rows + shared W", A, K * 2 + 128, W, K * 2 + 128, nk, 3, out);
    return 0;
}
