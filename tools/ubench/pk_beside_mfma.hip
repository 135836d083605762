// Do packed-f32 VALU instructions of one wave compute correctly while waves of ANOTHER kernel issue MFMAs on the same SIMD?
// (debug tool, not part of the product; background: DESIGN.md section 5d - the decoder's sampling kernel gave different bits in
// lanes 48-63 of a few waves when it ran beside the other launch chain's GEMM / attention kernels, and stopped doing so when it
// was compiled without v_pk_*_f32.)
//
// victim<F>: every lane runs a dependent chain of one instruction form on lane-dependent values, no memory traffic:
//   0 v_fma_f32 (two of them, the control)        1 v_pk_fma_f32, three VGPR pairs
//   2 v_pk_fma_f32 op_sel_hi:[0,1,1] (src0's low half for both lanes of the pair - the form the compiler emits for w * v8 + acc)
//   3 v_pk_mul_f32 with an SGPR pair              4 v_pk_fma_f32 with an inline constant, op_sel_hi:[1,0,1]
// aggressor<A>: MFMA chains, 4 independent accumulator tiles: 0 = 32x32x16 f16 into AGPRs, 1 = the same into VGPRs,
//   2 = 16x16x32 f16 into VGPRs, 3 = no MFMA (v_fma_f32 chains: a VALU-only neighbour).
// The victim runs alone (reference bits), then three times on a second stream while the aggressor runs on the first one.
//   hipcc --offload-arch=gfx950 -O3 pk_beside_mfma.hip -o pk_beside_mfma && ./pk_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int F>
__global__ __launch_bounds__(256) void victim(float* out, int iters, float s_lo, float s_hi) {
    const int lane = threadIdx.x & 63;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    f32x2 acc = {1.0f + lane * 0.01f, 2.0f - lane * 0.01f};
    const f32x2 a = {0.5f + (tid & 7) * 0.01f, 0.25f + (tid & 3) * 0.02f};
    const f32x2 b = {0.125f * (lane & 3), 0.3f + 0.001f * (lane >> 4)};
    f32x2 sp = {s_lo, s_hi};
    for (int i = 0; i < iters; ++i) {
        if (F == 0) {
            float x = acc[0], y = acc[1];
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a[0]), "v"(b[0]));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(a[1]), "v"(b[1]));
            acc[0] = x; acc[1] = y;
        } else if (F == 1) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
        } else if (F == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
        } else if (F == 3) {
            asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc) : "s"(sp));
        } else {
            asm volatile("v_pk_fma_f32 %0, %0, 0.5, %1 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(b));
        }
    }
    out[tid * 2] = acc[0]; out[tid * 2 + 1] = acc[1];
}

template <int A>
__global__ __launch_bounds__(256) void aggressor(float* sink, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
    float keep = 0.f;
    if (A == 0) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c2) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c3) : "v"(a), "v"(b));
        }
        keep = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (A == 1) {
        f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
        }
        keep = c0[0] + c1[1] + c2[2] + c3[3];
    } else if (A == 2) {
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
        }
        keep = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
        const float m = 0.999f, ad = 0.001f;
        for (int i = 0; i < iters * 8; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(m), "v"(ad));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(m), "v"(ad));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(m), "v"(ad));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(m), "v"(ad));
        }
        keep = x0 + x1 + x2 + x3;
    }
    if (keep == 123.456f) sink[0] = keep;
}

template <int F> void launch_victim(float* out, int grid, int iters, hipStream_t s) {
    hipLaunchKernelGGL(victim<F>, dim3(grid), dim3(256), 0, s, out, iters, 1.0009765625f, 0.9990234375f);
}
void launch_victim(int f, float* out, int grid, int iters, hipStream_t s) {
    switch (f) {
        case 0: launch_victim<0>(out, grid, iters, s); break;
        case 1: launch_victim<1>(out, grid, iters, s); break;
        case 2: launch_victim<2>(out, grid, iters, s); break;
        case 3: launch_victim<3>(out, grid, iters, s); break;
        default: launch_victim<4>(out, grid, iters, s); break;
    }
}
void launch_aggressor(int a, float* sink, int grid, int iters, hipStream_t s) {
    switch (a) {
        case 0: hipLaunchKernelGGL(aggressor<0>, dim3(grid), dim3(256), 0, s, sink, iters); break;
        case 1: hipLaunchKernelGGL(aggressor<1>, dim3(grid), dim3(256), 0, s, sink, iters); break;
        case 2: hipLaunchKernelGGL(aggressor<2>, dim3(grid), dim3(256), 0, s, sink, iters * 4); break;
        default: hipLaunchKernelGGL(aggressor<3>, dim3(grid), dim3(256), 0, s, sink, iters); break;
    }
}

int main(int argc, char** argv) {
    const int vgrid = 2048, viters = argc > 1 ? atoi(argv[1]) : 4000, agrid = 512, aiters = argc > 2 ? atoi(argv[2]) : 6000;
    const int trials = argc > 3 ? atoi(argv[3]) : 12;
    const long n = (long)vgrid * 256 * 2;
    float *ref, *out[3], *sink;
    hipMalloc(&ref, n * 4); hipMalloc(&sink, 64);
    for (int i = 0; i < 3; ++i) hipMalloc(&out[i], n * 4);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    std::vector<float> href(n), hout(n);
    const char* fn[] = {"v_fma_f32 x2 (control)", "v_pk_fma_f32 vgpr,vgpr,vgpr", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_mul_f32 sgpr pair", "v_pk_fma_f32 inline 0.5"};
    const char* an[] = {"32x32x16 MFMA -> AGPR", "32x32x16 MFMA -> VGPR", "16x16x32 MFMA -> VGPR", "v_fma_f32 only (no MFMA)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int a = 0; a < 4; ++a) {                       // how long the aggressor runs alone
        hipEventRecord(e0, s1); launch_aggressor(a, sink, agrid, aiters, s1); hipEventRecord(e1, s1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("aggressor %d (%s) alone: %.0f us\n", a, an[a], ms * 1e3);
    }
    for (int f = 0; f < 5; ++f) {
        hipEventRecord(e0, s2); launch_victim(f, ref, vgrid, viters, s2); hipEventRecord(e1, s2); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(href.data(), ref, n * 4, hipMemcpyDeviceToHost);
        launch_victim(f, out[0], vgrid, viters, s2); hipStreamSynchronize(s2);
        hipMemcpy(hout.data(), out[0], n * 4, hipMemcpyDeviceToHost);
        printf("victim %d (%s): %.0f us alone, repeat alone %s\n", f, fn[f], ms * 1e3, memcmp(href.data(), hout.data(), n * 4) ? "DIFFERS" : "identical");
        for (int a = 0; a < 4; ++a) {
            long bad_runs = 0, bad_elems = 0, q[4] = {0, 0, 0, 0}, bad_waves = 0;
            for (int t = 0; t < trials; ++t) {
                launch_aggressor(a, sink, agrid, aiters, s1);
                for (int i = 0; i < 3; ++i) launch_victim(f, out[i], vgrid, viters, s2);
                hipDeviceSynchronize();
                for (int i = 0; i < 3; ++i) {
                    hipMemcpy(hout.data(), out[i], n * 4, hipMemcpyDeviceToHost);
                    if (!memcmp(href.data(), hout.data(), n * 4)) continue;
                    ++bad_runs;
                    long lastw = -1;
                    for (long e = 0; e < n; ++e)
                        if (memcmp(&href[e], &hout[e], 4)) {
                            ++bad_elems; ++q[((e >> 1) & 63) >> 4];
                            if ((e >> 7) != lastw) { ++bad_waves; lastw = e >> 7; }
                        }
                }
            }
            printf("    beside %-26s: %3ld of %d runs differ; %ld elements in %ld waves; by lane quarter %ld %ld %ld %ld\n", an[a], bad_runs,
                   trials * 3, bad_elems, bad_waves, q[0], q[1], q[2], q[3]);
        }
    }
    return 0;
}
