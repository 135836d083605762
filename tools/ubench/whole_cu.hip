// How many workgroups that need a CU to themselves run at the same time? (box diagnostic, tools/box_info.sh)
// Two kinds: (a) 4 waves x 512 registers (the ViT block kernel's footprint) + 108 KB of LDS, (b) 160 KB of LDS (the large-tile GEMM's).
// Every workgroup spins for ~100 us; N workgroups take ceil(N / (CUs that can host one)) x 100 us. On a free MI355X all of 256 run
// at once; profiles/r3f_box_spread.txt is about boxes where kernels of this kind ran 9-95 % slower and nothing else did.
//   hipcc --offload-arch=gfx950 -O3 whole_cu.hip -o whole_cu && ./whole_cu
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool REGS>
__global__ __launch_bounds__(256, 1) void spin(unsigned long long ticks, int* sink) {
    extern __shared__ char lds[];
    if (REGS) asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, v255" ::: "v255", "a255");      // allocates all 512 registers
    lds[threadIdx.x] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[threadIdx.x] == 123) sink[0] = 1;
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)spin<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)spin<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned long long ticks = 10000;        // 100 MHz counter: 100 us
    for (int kind = 0; kind < 2; ++kind) {
        printf("%s:", kind == 0 ? "4 waves x 512 registers + 108 KB LDS" : "160 KB LDS");
        for (int n : {64, 128, 192, 224, 240, 248, 256, 264, 512}) {
            float best = 1e9f;
            for (int it = 0; it < 3; ++it) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(spin<true>, dim3(n), dim3(256), 108 * 1024, 0, ticks, sink);
                else hipLaunchKernelGGL(spin<false>, dim3(n), dim3(256), 160 * 1024, 0, ticks, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("  %d wg %.0f us", n, best * 1e3);
        }
        printf("\n");
    }
    return 0;
}
