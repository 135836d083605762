// Issue cost of the VALU / transcendental instructions the softmax and GELU inner loops are made of (gfx950): cycles per
// wave-instruction at 1 / 2 / 4 waves per SIMD, 8 independent registers round-robin (throughput, not latency).
//   hipcc --offload-arch=gfx950 -O3 valu.hip -o valu
#include <hip/hip_runtime.h>
#include <stdio.h>

#define R8(OP)  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define DEF(NAME, ASM)                                                                                                   \
    __global__ __launch_bounds__(1024) void NAME(float* out, unsigned long long* cyc, int iters) {                       \
        float v0 = threadIdx.x * 0.001f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6,    \
              v7 = v0 + 7, k = 0.999f;                                                                                   \
        __syncthreads();                                                                                                 \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                      \
        for (int it = 0; it < iters; ++it) {                                                                             \
            asm volatile(ASM ASM ASM ASM                                                                                 \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k));      \
        }                                                                                                                \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                              \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;                                \
    }

#define E8(s0, s1, s2, s3, s4, s5, s6, s7) s0 s1 s2 s3 s4 s5 s6 s7
DEF(k_exp32, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
DEF(k_exp16, "v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n")
DEF(k_rcp32, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
DEF(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
DEF(k_fma, "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
DEF(k_max3, "v_max3_f32 %0, %0, %1, %8\n v_max3_f32 %1, %1, %2, %8\n v_max3_f32 %2, %2, %3, %8\n v_max3_f32 %3, %3, %4, %8\n v_max3_f32 %4, %4, %5, %8\n v_max3_f32 %5, %5, %6, %8\n v_max3_f32 %6, %6, %7, %8\n v_max3_f32 %7, %7, %0, %8\n")
DEF(k_cvtpk, "v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %4\n v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %5, %5, %6\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %7, %7, %0\n")
DEF(k_cvtpkbf, "v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n")
DEF(k_dot2c, "v_dot2c_f32_f16 %0, %1, %8\n v_dot2c_f32_f16 %1, %2, %8\n v_dot2c_f32_f16 %2, %3, %8\n v_dot2c_f32_f16 %3, %4, %8\n v_dot2c_f32_f16 %4, %5, %8\n v_dot2c_f32_f16 %5, %6, %8\n v_dot2c_f32_f16 %6, %7, %8\n v_dot2c_f32_f16 %7, %0, %8\n")
DEF(k_pkmulf16, "v_pk_mul_f16 %0, %0, %8\n v_pk_mul_f16 %1, %1, %8\n v_pk_mul_f16 %2, %2, %8\n v_pk_mul_f16 %3, %3, %8\n v_pk_mul_f16 %4, %4, %8\n v_pk_mul_f16 %5, %5, %8\n v_pk_mul_f16 %6, %6, %8\n v_pk_mul_f16 %7, %7, %8\n")
DEF(k_pkfmaf16, "v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8\n")
DEF(k_or3, "v_or3_b32 %0, %0, %1, %8\n v_or3_b32 %1, %1, %2, %8\n v_or3_b32 %2, %2, %3, %8\n v_or3_b32 %3, %3, %4, %8\n v_or3_b32 %4, %4, %5, %8\n v_or3_b32 %5, %5, %6, %8\n v_or3_b32 %6, %6, %7, %8\n v_or3_b32 %7, %7, %0, %8\n")
DEF(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n")
DEF(k_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %1, %1, %2\n v_cvt_pkrtz_f16_f32 %2, %2, %3\n v_cvt_pkrtz_f16_f32 %3, %3, %4\n v_cvt_pkrtz_f16_f32 %4, %4, %5\n v_cvt_pkrtz_f16_f32 %5, %5, %6\n v_cvt_pkrtz_f16_f32 %6, %6, %7\n v_cvt_pkrtz_f16_f32 %7, %7, %0\n")
DEF(k_max2, "v_max_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_max_f32 %3, %3, %4\n v_max_f32 %4, %4, %5\n v_max_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_max_f32 %7, %7, %0\n")
DEF(k_or2, "v_or_b32 %0, %0, %1\n v_or_b32 %1, %1, %2\n v_or_b32 %2, %2, %3\n v_or_b32 %3, %3, %4\n v_or_b32 %4, %4, %5\n v_or_b32 %5, %5, %6\n v_or_b32 %6, %6, %7\n v_or_b32 %7, %7, %0\n")
DEF(k_sub, "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n")
DEF(k_ldexp, "v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8\n")

typedef void (*kfn)(float*, unsigned long long*, int);
static void run(const char* what, kfn f) {
    float* out; unsigned long long* cyc; unsigned long long h[16];
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 128);
    const int iters = 400;
    printf("%-22s", what);
    for (int threads : {256, 512, 1024}) {
        (void)hipMemset(cyc, 0, 128);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(f, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        double mx = 0; for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
        // cycles per wave-instruction on the SIMD = elapsed / (iters * 32 instr) / waves-per-SIMD
        printf("  %dw/SIMD: %5.2f", threads / 256, mx / (iters * 32.0) / (threads / 256));
    }
    printf("   cycles per wave-instruction (SIMD throughput)\n");
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run("v_exp_f32", k_exp32); run("v_exp_f16", k_exp16); run("v_rcp_f32", k_rcp32); run("v_mul_f32", k_mul);
    run("v_fma_f32", k_fma); run("v_max3_f32", k_max3); run("v_cvt_pk_f16_f32", k_cvtpk); run("v_cvt_pk_bf16_f32", k_cvtpkbf);
    run("v_dot2c_f32_f16", k_dot2c); run("v_pk_mul_f16", k_pkmulf16); run("v_pk_fma_f16", k_pkfmaf16); run("v_or3_b32", k_or3);
    run("v_permlane32_swap", k_swap32); run("v_ldexp_f32", k_ldexp);
    run("v_cvt_pkrtz_f16_f32", k_pkrtz); run("v_max_f32", k_max2); run("v_or_b32", k_or2); run("v_sub_f32", k_sub);
    return 0;
}
