// Store-pattern micro-benchmark for the q / k / v^T stores of the block kernel (tuning tool, not part of the product).
// Every wave writes R tiles of 2 KB (32 token rows of 64 bytes = one head's q rows of a 32-token tile). Pattern 0: the kernel's
// pattern - two instructions per tile, lane (j, h) writes 16 bytes at row j, chunk 2 i + h (32-byte pieces with 32-byte gaps per
// instruction). Pattern 1: two instructions per tile, each 1 KB contiguous (lane l at l * 16). Pattern 2: pattern 0 with 8-byte
// stores (the round-2 kernel). Prints GB/s.   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int PAT>
__global__ __launch_bounds__(256) void k(char* buf, int R, long stride_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const long w = (long)blockIdx.x * 4 + wave;
    const u32x4 v = {(unsigned)lane, (unsigned)w, 3u, 4u};
    for (int r = 0; r < R; ++r) {
        char* base = buf + ((long)r * stride_tiles + w) * 2048;           // tiles of one "piece" are neighbours across waves
        if (PAT == 0) {
            *(u32x4*)(base + j * 64 + (0 + h) * 16) = v;
            *(u32x4*)(base + j * 64 + (2 + h) * 16) = v;
        } else if (PAT == 1) {
            *(u32x4*)(base + lane * 16) = v;
            *(u32x4*)(base + 1024 + lane * 16) = v;
        } else {
            const u32x2 v2 = {v[0], v[1]};
#pragma unroll
            for (int c = 0; c < 4; ++c) *(u32x2*)(base + j * 64 + c * 16 + h * 8) = v2;
        }
    }
}
int main() {
    const int grid = 256, R = 58;                  // 1024 waves x 58 tiles x 2 KB = 118.8 MB (twice the kernel's 59 MB)
    const long nt = (long)grid * 4;
    char* buf; hipMalloc(&buf, nt * R * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 3; ++pat) {
        float best = 1e9f;
        for (int it = 0; it < 8; ++it) {
            hipEventRecord(e0);
            if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, buf, R, nt);
            else if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, buf, R, nt);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, buf, R, nt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 2 && ms < best) best = ms;
        }
        printf("pattern %d: %.1f us for %.1f MB -> %.0f GB/s\n", pat, best * 1e3, nt * R * 2048 / 1e6, nt * R * 2048 / (best * 1e-3) / 1e9);
    }
    return 0;
}
