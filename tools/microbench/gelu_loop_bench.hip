// Tuning tool (not part of the product): the hidden loop of vitblock_kernel<f16, 192, NH = 1, WPC = 2, G16> rebuilt in isolation - 24 MFMA slots per iteration
// (12 on six accumulators in turn = fc2, 12 on ONE accumulator = fc1), the packed-f16 GELU's 16 layer ticks handed out over the slots exactly as the kernel
// does - to find which ingredient keeps the ticks from hiding behind the matrix instructions (profiles/r6e_*: the ticks cost 0.41 us per iteration in
// the kernel, where the microbenchmark of single instruction kinds says five plain or three transcendental instructions per gap are free).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_timing/gelu_loop_bench tools/microbench/gelu_loop_bench.hip
// VARIANT bits: 1 = fc1 MFMAs on four accumulators in turn instead of one; 2 = a ds_read_b128 per slot (8 ahead, lgkmcnt-counted); 4 = layer 0 reads its
// values from accumulator registers (v_accvgpr_read); 8 = no ticks at all; 16 = transcendental ticks in halves (4 + 4 over two slots); 32 = no MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
#include <utility>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define ITERS 2048

template <typename F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ unsigned pack2(float a, float b) { const f32x2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2)); }
__device__ __forceinline__ unsigned g_mul(unsigned a, unsigned b) { unsigned r; asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned g_fma(unsigned a, unsigned c1s, unsigned c0v) { unsigned r; asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c1s), "v"(c0v)); return r; }
__device__ __forceinline__ unsigned g_add(unsigned a, unsigned ones) { unsigned r; asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(ones)); return r; }
#define TRANS4(OP)                                                                                                     \
    asm volatile(OP " %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t" OP " %1, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t" \
                 OP " %2, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t" OP " %3, %3 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t" \
                 OP " %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t" OP " %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t" \
                 OP " %2, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t" OP " %3, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t" \
                 "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define TRANS4H(OP, W)                                                                                                 \
    asm volatile(OP " %0, %0 dst_sel:" W " dst_unused:UNUSED_PRESERVE src0_sel:" W "\n\t" OP " %1, %1 dst_sel:" W " dst_unused:UNUSED_PRESERVE src0_sel:" W "\n\t" \
                 OP " %2, %2 dst_sel:" W " dst_unused:UNUSED_PRESERVE src0_sel:" W "\n\t" OP " %3, %3 dst_sel:" W " dst_unused:UNUSED_PRESERVE src0_sel:" W "\n\t" \
                 "s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

template <int V>
__global__ __launch_bounds__(256, 2) void gl(unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63;
    f16x8 bx;
    for (int i = 0; i < 8; ++i) bx[i] = (_Float16)(0.002f * (lane - i));
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float*)lds)[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc2[6] = {}, acc1[2][4] = {};
    unsigned hf[2][2][4] = {};
    unsigned c0, c1, one;
    asm volatile("v_mov_b32 %0, 0xc09ec09e" : "=v"(c0));
    asm volatile("s_mov_b32 %0, 0xae68ae68" : "=s"(c1));
    asm volatile("s_mov_b32 %0, 0x3c003c00" : "=s"(one));
    f16x8 fr[8];
    for (int i = 0; i < 8; ++i) fr[i] = *(const f16x8*)(lds + i * 1024 + lane * 16);
    unsigned gx[4] = {}, gq[4] = {};
    constexpr bool TICKS = !(V & 8), HALF = (V & 16) != 0;
    constexpr int NL = HALF ? 10 : 8, TK = NL * 2, S = 24;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < ITERS; it += 2) {
        sfor<2>([&](auto cur_tag) {
            constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
            sfor<S>([&](auto m_tag) {
                constexpr int m = decltype(m_tag)::value;
                constexpr int t_lo = m * TK / S, t_hi = (m + 1) * TK / S;
                auto tick = [&](auto ti_tag) {
                    constexpr int ti = decltype(ti_tag)::value, grp = ti / NL, L0 = ti % NL, r0 = 8 * grp;
                    // HALF: layers 0 1 2 3 | 4a 4b | 5 | 6a 6b | 7  ->  L = logical layer, H = which half
                    constexpr int L = !HALF ? L0 : (L0 < 4 ? L0 : L0 == 4 || L0 == 5 ? 4 : L0 == 6 ? 5 : L0 == 7 || L0 == 8 ? 6 : 7);
                    constexpr int H = !HALF ? -1 : (L0 == 4 || L0 == 7 ? 0 : L0 == 5 || L0 == 8 ? 1 : -1);
                    if constexpr (L == 4) { unsigned &a = gq[0], &b = gq[1], &c = gq[2], &d = gq[3];
                        if constexpr (H < 0) TRANS4("v_exp_f16_sdwa"); else if constexpr (H == 0) TRANS4H("v_exp_f16_sdwa", "WORD_0"); else TRANS4H("v_exp_f16_sdwa", "WORD_1"); }
                    else if constexpr (L == 6) { unsigned &a = gq[0], &b = gq[1], &c = gq[2], &d = gq[3];
                        if constexpr (H < 0) TRANS4("v_rcp_f16_sdwa"); else if constexpr (H == 0) TRANS4H("v_rcp_f16_sdwa", "WORD_0"); else TRANS4H("v_rcp_f16_sdwa", "WORD_1"); }
                    else {
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            if constexpr (L == 0) {
                                if constexpr (V & 4) gx[d] = pack2(acc1[CUR][0][r0 + 2 * d], acc1[CUR][0][r0 + 2 * d + 1]);
                                else gx[d] = pack2(__builtin_bit_cast(float, gq[d]), __builtin_bit_cast(float, gx[d]));
                            }
                            else if constexpr (L == 1) gq[d] = g_mul(gx[d], gx[d]);
                            else if constexpr (L == 2) gq[d] = g_fma(gq[d], c1, c0);
                            else if constexpr (L == 3) gq[d] = g_mul(gx[d], gq[d]);
                            else if constexpr (L == 5) gq[d] = g_add(gq[d], one);
                            else gq[d] = g_mul(gx[d], gq[d]);
                        }
                    }
                    if constexpr (L == 0) asm volatile("" : "+v"(gx[0]), "+v"(gx[1]), "+v"(gx[2]), "+v"(gx[3]));
                    if constexpr (L == 7) { hf[CUR][grp][0] = gq[0]; hf[CUR][grp][1] = gq[1]; hf[CUR][grp][2] = gq[2]; hf[CUR][grp][3] = gq[3]; }
                };
                if constexpr (TICKS) {
                    if constexpr (t_lo < t_hi) tick(std::integral_constant<int, t_lo>{});
                    if constexpr (t_lo + 1 < t_hi) tick(std::integral_constant<int, t_lo + 1>{});
                }
                const f16x8 a = fr[m % 8];
                if constexpr (!(V & 32)) {
                    if constexpr (m < 12) {
                        f16x8 hb; for (int e = 0; e < 4; ++e) { const f16x2 p = __builtin_bit_cast(f16x2, hf[NXT][m / 6][e]); hb[2 * e] = p[0]; hb[2 * e + 1] = p[1]; }
                        acc2[m % 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, hb, acc2[m % 6], 0, 0, 0);
                    } else {
                        constexpr int w = (V & 1) ? (m & 3) : 0;
                        acc1[NXT][w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bx, acc1[NXT][w], 0, 0, 0);
                    }
                }
                if constexpr (V & 2) fr[m % 8] = *(const f16x8*)(lds + ((m + 8) % 24) * 1024 + lane * 16);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; if (blockIdx.x == 0) out[gridDim.x] = w1 - w0; }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 16; ++e) s += acc2[i][e];
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc1[b][i][e];
    for (int i = 0; i < 4; ++i) s += (float)gx[i] + (float)gq[i] + (float)hf[0][0][i] + (float)hf[1][1][i];
    for (int i = 0; i < 8; ++i) s += (float)fr[i][0];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int V> void run(unsigned long long* d_out, float* d_sink, int grid, const char* what) {
    std::vector<unsigned long long> h(grid + 1);
    double best = 1e30, mhz = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((gl<V>), dim3(grid), dim3(256), 0, 0, d_out, d_sink);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d_out, (grid + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const double v = (double)h[0] / ITERS;
        if (v < best) { best = v; mhz = (double)h[0] / ((double)h[grid] * 0.01); }
    }
    printf("variant %2d: %7.1f cycles per iteration of 24 MFMAs (%5.1f per MFMA)  %4.0f MHz   %s\n", V, best, best / 24.0, mhz, what);
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    unsigned long long* d_out; float* d_sink;
    hipMalloc(&d_out, (grid + 1) * sizeof(unsigned long long)); hipMalloc(&d_sink, 1024);
    printf("grid %d x 256 threads\n", grid);
    run<8>(d_out, d_sink, grid, "MFMAs only (fc1 on one accumulator)");
    run<9>(d_out, d_sink, grid, "MFMAs only, fc1 on four accumulators");
    run<10>(d_out, d_sink, grid, "MFMAs + fragment reads");
    run<0>(d_out, d_sink, grid, "ticks (values from VGPRs), no fragment reads");
    run<1>(d_out, d_sink, grid, "ticks, fc1 on four accumulators");
    run<4>(d_out, d_sink, grid, "ticks with accumulator reads in layer 0");
    run<6>(d_out, d_sink, grid, "ticks with accumulator reads + fragment reads (= the kernel's loop without DMA)");
    run<7>(d_out, d_sink, grid, "the same, fc1 on four accumulators");
    run<22>(d_out, d_sink, grid, "kernel's loop, transcendental ticks in halves");
    run<23>(d_out, d_sink, grid, "the same, fc1 on four accumulators");
    run<38>(d_out, d_sink, grid, "ticks + accumulator reads + fragment reads, NO MFMAs");
    return 0;
}
