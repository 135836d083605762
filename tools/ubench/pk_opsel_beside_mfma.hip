// Does `v_pk_fma_f32 ... op_sel:[0,1,0] op_sel_hi:[1,0,0]` (src1 with its halves CROSSED: the low result reads the pair's high
// register and vice versa) compute correctly while waves of ANOTHER kernel issue MFMAs on the same SIMD?  (debug tool)
//
// Background (profiles/r4a_msda_isa_bisect.txt): tools/msda_isa_probe.py ran the sampling kernel's own
// assembly with ONE packed instruction left packed at a time beside the other launch chain's attention / GEMM kernels: the only
// instruction whose packed form gives wrong results (lanes 48-63 of a few waves per launch) is this one; the crossed v_pk_add_f32,
// every straight v_pk_{fma,mul}_f32 and the SGPR-pair forms never do. Round 3's ubench (pk_beside_mfma.hip) did not contain the form,
// and its victims were dependent contractions (a transient error decays). Here every iteration computes the packed form on fresh,
// iteration-dependent inputs and compares it - in the same thread - with two scalar instructions; mismatches are counted per lane.
//   hipcc --offload-arch=gfx950 -O3 pk_opsel_beside_mfma.hip -o pk_opsel_beside_mfma && ./pk_opsel_beside_mfma [iters] [trials]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// forms: 0 crossed src1 + inline constant (THE instruction)   1 crossed src1, VGPR src2   2 straight, inline constant
//        3 crossed v_pk_mul_f32   4 crossed v_pk_add_f32 with neg (clean in the kernel)   5 straight v_pk_fma_f32, all VGPR
//        6 crossed src0 + inline constant
template <int F>
__global__ __launch_bounds__(256) void victim(unsigned* bad, int iters) {
    const int lane = threadIdx.x & 63;
    const long tid = (long)blockIdx.x * 256 + threadIdx.x;
    f32x2 x = {0.37f + lane * 0.013f, 0.81f - lane * 0.007f};
    const f32x2 b = {40.f + (tid & 7), 40.f - (tid & 3)};
    const f32x2 c = {0.125f * (lane & 3), 0.3f + 0.001f * (lane >> 4)};
    unsigned nbad = 0;
    for (int i = 0; i < iters; ++i) {
        f32x2 r; float e0, e1;
        if (F == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, -0.5 op_sel:[0,1,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(e0) : "v"(x[0]), "v"(b[1]));
            asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(e1) : "v"(x[1]), "v"(b[0]));
        } else if (F == 1) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x[0]), "v"(b[1]), "v"(c[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x[1]), "v"(b[0]), "v"(c[1]));
        } else if (F == 2) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, -0.5 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(e0) : "v"(x[0]), "v"(b[0]));
            asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(e1) : "v"(x[1]), "v"(b[1]));
        } else if (F == 3) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(b));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(x[0]), "v"(b[1]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(x[1]), "v"(b[0]));
        } else if (F == 4) {
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(b));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e0) : "v"(x[1]), "v"(b[0]));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e1) : "v"(x[0]), "v"(b[1]));
        } else if (F == 5) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(x[0]), "v"(b[0]), "v"(c[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(x[1]), "v"(b[1]), "v"(c[1]));
        } else {
            asm volatile("v_pk_fma_f32 %0, %1, %2, -0.5 op_sel:[1,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(e0) : "v"(x[1]), "v"(b[0]));
            asm volatile("v_fma_f32 %0, %1, %2, -0.5" : "=v"(e1) : "v"(x[0]), "v"(b[1]));
        }
        nbad += (__float_as_uint(r[0]) != __float_as_uint(e0)) | (__float_as_uint(r[1]) != __float_as_uint(e1));
        x[0] = x[0] * 0.999f + 0.0007f; x[1] = x[1] * 1.001f - 0.0004f;        // fresh inputs every iteration (plain VALU)
    }
    bad[tid] = nbad;
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
        for (int i = 0; i < iters * 4; ++i) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
        }
        keep = c0[0] + c1[1] + c2[2] + c3[3];
    } else {        // MFMA results consumed by VALU code, exp2 in between: the shape of the attention kernel without LDS (the strongest trigger)
        f32x4 c0 = {}, c1 = {};
        float m = 0.f;
        for (int i = 0; i < iters * 2; ++i) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            m = fmaxf(m, __builtin_amdgcn_exp2f(c0[0] * 1e-6f) + __builtin_amdgcn_exp2f(c1[1] * 1e-6f));
            c0[1] *= 0.5f; c1[2] *= 0.5f;
        }
        keep = c0[0] + c1[1] + m;
    }
    if (keep == 123.456f) sink[0] = keep;
}

template <int F> void lv(unsigned* bad, int grid, int iters, hipStream_t s) { hipLaunchKernelGGL(victim<F>, dim3(grid), dim3(256), 0, s, bad, iters); }
void launch_victim(int f, unsigned* bad, int grid, int iters, hipStream_t s) {
    switch (f) {
        case 0: lv<0>(bad, grid, iters, s); break; case 1: lv<1>(bad, grid, iters, s); break; case 2: lv<2>(bad, grid, iters, s); break;
        case 3: lv<3>(bad, grid, iters, s); break; case 4: lv<4>(bad, grid, iters, s); break; case 5: lv<5>(bad, grid, iters, s); break;
        default: lv<6>(bad, grid, iters, s); break;
    }
}
void launch_aggressor(int a, float* sink, int grid, int iters, hipStream_t s) {
    switch (a) {
        case 0: hipLaunchKernelGGL(aggressor<0>, dim3(grid), dim3(256), 0, s, sink, iters); break;
        case 1: hipLaunchKernelGGL(aggressor<1>, dim3(grid), dim3(256), 0, s, sink, iters); break;
        case 2: hipLaunchKernelGGL(aggressor<2>, dim3(grid), dim3(256), 0, s, sink, iters); break;
        default: hipLaunchKernelGGL(aggressor<3>, dim3(grid), dim3(256), 0, s, sink, iters); break;
    }
}

int main(int argc, char** argv) {
    const int vgrid = 2048, viters = argc > 1 ? atoi(argv[1]) : 2000, agrid = 512, aiters = 6000;
    const int trials = argc > 2 ? atoi(argv[2]) : 8;
    const long n = (long)vgrid * 256;
    unsigned* bad; float* sink;
    hipMalloc(&bad, n * 4); hipMalloc(&sink, 64);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    std::vector<unsigned> hb(n);
    const char* fn[] = {"v_pk_fma_f32 x, b, -0.5 op_sel:[0,1,0] op_sel_hi:[1,0,0] (crossed src1)", "v_pk_fma_f32 crossed src1, VGPR src2",
                        "v_pk_fma_f32 straight, inline -0.5", "v_pk_mul_f32 crossed src1", "v_pk_add_f32 crossed src0 + neg (clean in the kernel)",
                        "v_pk_fma_f32 straight, all VGPR", "v_pk_fma_f32 crossed src0, inline -0.5"};
    const char* an[] = {"none", "32x32x16 MFMA -> AGPR", "32x32x16 MFMA -> VGPR", "16x16x32 MFMA -> VGPR", "16x16x32 MFMA + exp2 + VALU"};
    for (int f = 0; f < 7; ++f) {
        printf("victim %d: %s\n", f, fn[f]);
        for (int a = -1; a < 4; ++a) {
            long bad_launches = 0, bad_execs = 0, q[4] = {0, 0, 0, 0}, bad_waves = 0;
            for (int t = 0; t < trials; ++t) {
                if (a >= 0) launch_aggressor(a, sink, agrid, aiters, s1);
                for (int i = 0; i < 3; ++i) {
                    launch_victim(f, bad, vgrid, viters, s2);
                    hipStreamSynchronize(s2);
                    hipMemcpy(hb.data(), bad, n * 4, hipMemcpyDeviceToHost);
                    long nb = 0, lastw = -1;
                    for (long e = 0; e < n; ++e)
                        if (hb[e]) { nb += hb[e]; q[(e & 63) >> 4] += hb[e]; if ((e >> 6) != lastw) { ++bad_waves; lastw = e >> 6; } }
                    bad_execs += nb; bad_launches += nb != 0;
                }
                hipDeviceSynchronize();
            }
            printf("    beside %-28s: %3ld of %d launches with mismatches; %ld wrong executions of %.3g in %ld waves; by lane quarter %ld %ld %ld %ld\n",
                   an[a + 1], bad_launches, trials * 3, bad_execs, (double)n * viters * trials * 3, bad_waves, q[0], q[1], q[2], q[3]);
        }
    }
    return 0;
}
