// CU ingress microbenchmark (tuning tool): bytes per clock one CU can pull from an L2-resident buffer, through plain
// 16-byte global loads and through LDS-DMA (global_load_lds_dwordx4).  hipcc --offload-arch=gfx950 -O3 ingress.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int UNROLL>
__global__ __launch_bounds__(512) void k(const f32x4* __restrict__ src, int region_vec, float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[8 * 64 * UNROLL];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // every workgroup walks the same region (like weights shared by all CUs), each wave its own 1 KB pieces
    int pos = (wave * 64 * UNROLL + lane) % region_vec;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = src[(pos + u * 64) % region_vec];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        } else {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const f32x4* p = src + (pos + u * 64) % region_vec;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * UNROLL + u) * 64), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        pos = (pos + 8 * 64 * UNROLL) % region_vec;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 1) acc = lds[threadIdx.x];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE, int UNROLL>
void run(const f32x4* src, int region_bytes, int nblocks, const char* what) {
    float* out; unsigned long long* cyc; unsigned long long h = 0;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(nblocks), dim3(512), 0, 0, src, region_bytes / 16, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double bytes = (double)iters * 8 * 64 * UNROLL * 16;
    printf("%-34s region %6d KB  WGs %3d  unroll %d: %.1f B/clk/CU\n", what, region_bytes / 1024, nblocks, UNROLL, bytes / (double)h);
    hipFree(out); hipFree(cyc);
}

// strided pieces: lane l fetches 16 B at row (l >> 2) * pitch + (l & 3) * 16 (+ 64 B per k-step): a 16-row x 64-byte GEMM tile slice
template <int MODE, int SEG>
__global__ __launch_bounds__(512) void ks(const char* __restrict__ src, long pitch, long region, float* out, unsigned long long* cyc, int iters) {
    __shared__ f32x4 lds[8 * 64 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int LPR = SEG / 16;                   // lanes per row
    const long lane_off = (long)(lane / LPR) * pitch + (lane % LPR) * 16;
    long base = ((long)wave * (64 / LPR) * 4) * pitch;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    long kofs = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const char* p = src + (base + (long)u * (64 / LPR) * pitch + lane_off + kofs) % region;
            if (MODE == 0) acc += *(const f32x4*)p;
            else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                  (__attribute__((address_space(3))) void*)(lds + (wave * 4 + u) * 64), 16, 0, 0);
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        kofs += SEG;
        if (kofs >= pitch) { kofs = 0; base += 8L * (64 / LPR) * 4 * pitch; }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 1) acc = lds[threadIdx.x];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int SEG>
void runs(const char* src, long pitch, long region, const char* what) {
    float* out; unsigned long long* cyc; unsigned long long h = 0;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((ks<MODE, SEG>), dim3(256), dim3(512), 0, 0, src, pitch, region, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s seg %3d B pitch %5ld region %5ld KB: %.1f B/clk/CU\n", what, SEG, pitch, region / 1024, (double)iters * 8 * 4 * 1024 / (double)h);
    hipFree(out); hipFree(cyc);
}

int main() {
    f32x4* src; hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20);
    for (int region : {64 << 10, 512 << 10, 4 << 20, 48 << 20}) {
        run<0, 4>(src, region, 256, "global_load_dwordx4 -> VGPR");
        run<1, 4>(src, region, 256, "global_load_lds_dwordx4 (DMA)");
    }
    run<0, 8>(src, 512 << 10, 256, "global_load_dwordx4 -> VGPR");
    run<1, 8>(src, 512 << 10, 256, "global_load_lds_dwordx4 (DMA)");
    run<0, 4>(src, 512 << 10, 1, "global_load_dwordx4, ONE CU busy");
    run<1, 4>(src, 512 << 10, 1, "DMA, ONE CU busy");
    run<0, 4>(src, 512 << 10, 32, "global_load_dwordx4, 32 CUs busy");
    run<1, 4>(src, 512 << 10, 32, "DMA, 32 CUs busy");
    for (long pitch : {256L, 384L, 2304L}) {
        runs<0, 64>((const char*)src, pitch, 2L << 20, "strided -> VGPR");
        runs<1, 64>((const char*)src, pitch, 2L << 20, "strided DMA");
        runs<1, 128>((const char*)src, pitch, 2L << 20, "strided DMA");
        runs<1, 256>((const char*)src, pitch, 2L << 20, "strided DMA");
    }
    return 0;
}
