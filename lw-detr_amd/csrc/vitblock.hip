// Fused ViT block tail for gfx950, 16-bit types, C in {192, 384} (round 3; replaces mlp_kernel at BASELINE batch sizes):
//
//   x1 = x + gamma1 * (att Wp^T + bp)                       attention output projection   (vit.py:138, :206-216)
//   x  = x1 + gamma2 * (fc2(GELU(fc1(LN2(x1)))) + b2)       MLP                           (vit.py:217-218, timm Mlp)
//   q, k, v^T = heads(LN1'(x) Wqkv'^T + b')                 norm1 + QKV of the NEXT block (vit.py:199, :123-130), optional
//
// Why it looks the way it does (measured on the round-2 kernel, profiles/r2*):
//   * the matrix pipe, the VALU (GELU: 2 transcendentals + 7 plain instructions per hidden value) and the LDS -> register
//     fragment returns all cost about the same per hidden chunk at C = 192, and the old kernel ran them one after the other
//     (all 8 waves in lockstep: MFMA phase, GELU phase, barrier). Here ONE wave per SIMD (4 per workgroup, 512 registers)
//     owns 64 tokens (C = 192) / 32 tokens (C = 384) and runs a software pipeline: while GELU of chunk k runs on the VALU,
//     the same instruction stream issues fc2 of chunk k-1 and fc1 of chunk k+1 (one 32x32x16 MFMA every ~7 VALU
//     instructions, pinned with sched_barrier). Twice the tokens per wave = half the weight-fragment traffic per MFMA.
//   * every weight of the block (Wp, W1, W2, Wqkv') is packed ON THE HOST into one stream of 32-row pieces, each piece a
//     sequence of 1 KB MFMA fragments in lane order (lane l's 16 bytes at l * 16): the DMA is linear, a fragment read is
//     `base + lane * 16 + immediate`, conflict-free by construction, no swizzle arithmetic, no padded rows. The stream
//     runs through ONE ring of NSLOT pieces with counted s_waitcnt vmcnt + raw s_barrier per step (2 pieces), several
//     steps ahead - the old kernel restarted a 2-deep double buffer (and paid a DMA round trip) per phase.
//   * the residual stream never sits in LDS or goes back to HBM between the projection and the epilogue: the fc2
//     accumulators start at x1 / gamma2 + b2 and the epilogue is one multiply (|x1 / gamma2| only costs f32 round-off of
//     x1, far below the 16-bit output grid; the host refuses gamma2 entries that are zero / denormal).
// Fragment conventions (32x32x16 MFMA, wave64): A lane (i = l & 31, h = l >> 5) holds A[i][8h .. 8h+7]; B the same with
// j = l & 31; D register 4 b + e of lane (j, h) is D[8 b + 4 h + e][j]. An accumulator tile therefore IS the B operand
// of the next contraction over its rows, with k-slot (step 2n + beta, half h, s = 4 b' + e) <-> row 32 n + 16 beta + 8 b' +
// 4 h + e: the host permutes the weight columns accordingly (lwdetr_amd/kernels.py:pack_vit_block).
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

// Phase timing for kernel tuning (tools/vitblock_timing.py builds a private copy with -DLWDETR_VB_TIMING; never in the product
// library): s_memrealtime stamps (10 ns ticks) of every wave of two workgroups at the phase boundaries, plus the time spent in
// the ring waits and barriers of the hidden loop.
// Ablation builds for tuning (-DLWDETR_VB_ABLATE=bits, results are WRONG; never in the product): 1 = no GELU ticks in the hidden
// loop, 2 = no MFMAs there, 4 = no weight DMA after the prologue, 8 = no fragment reads in the hidden loop.
#ifndef LWDETR_VB_ABLATE
#define LWDETR_VB_ABLATE 0
#endif
#ifdef LWDETR_VB_TIMING
__device__ unsigned long long g_vb_timing[2][4][16];
// (slots 12 / 15: s_memtime = shader clocks at the first / last stamp - the clock the kernel actually ran at)
#define VB_TS(i) do { if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane == 0) { g_vb_timing[blockIdx.x != 0][wave][i] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 0) g_vb_timing[blockIdx.x != 0][wave][12] = __builtin_amdgcn_s_memtime(); if ((i) == 11) g_vb_timing[blockIdx.x != 0][wave][15] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" int lwdetr_debug_vb_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vb_timing), sizeof(g_vb_timing)) == hipSuccess ? 0 : 1;
}
#else
#define VB_TS(i) do {} while (0)
#endif

namespace {

template <typename T> struct Mma32;
template <> struct Mma32<f16> {
    static __device__ __forceinline__ f32x16 k16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32<bf16> {
    static __device__ __forceinline__ f32x16 k16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

// compile-time loop: f(integral_constant<int, 0>{}) ... f(integral_constant<int, N-1>{}) (the hidden loop's slots must be
// constants for `if constexpr`; a 48-iteration `#pragma unroll` body with run-time-looking branches exceeds hipcc's threshold)
template <typename F, int... I> __device__ __forceinline__ void vb_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void vb_static_for(F&& f) {
    vb_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <typename T> struct Pk;
template <> struct Pk<f16> { typedef f16x2 v2; };
template <> struct Pk<bf16> { typedef bf16x2 v2; };
// two f32 -> one packed dword of T (round to nearest even: v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32)
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, typename Pk<T>::v2));
}
// Accumulator registers 8 jb .. 8 jb + 7 of a tile are rows 16 jb + 4 h + {0..3} (a) and 16 jb + 8 + 4 h + {0..3} (b) of column
// j. Exchanging a's upper half-wave with b's lower one (v_permlane32_swap) leaves every lane with 8 CONSECUTIVE rows
// 16 jb + 8 h .. + 7: one 16-byte store instead of two 8-byte ones (the store path is issue-bound: 12 us of epilogue measured
// with 8-byte pieces). a / b: the lane's own two packed pairs each.
__device__ __forceinline__ u32x4 vb_rows8(unsigned a0, unsigned a1, unsigned b0, unsigned b1) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct VbParams {
    void* x; long ldx;                 // (M, C) residual stream, updated in place
    const void* att; long ldatt;       // (M, C) attention output (heads concatenated)
    const void* wstream;               // packed weight pieces (see pack_vit_block)
    const float* vec;                  // packed f32 vectors: b1'[4C] | bp[C] | g1[C] | 1/g1[C] | b2[C] | 1/g2[C] | g2[C] | bqkv'[3C], padded
    void* out2; long ld2;              // optional copy of the new rows (ViT feature taps)
    float* stats_out;                  // optional (M, 2): mean, rstd of the new rows (eps_next)
    void* q; void* k; void* vt;        // QKV outputs of the next block
    long M;
    float eps, eps_next, qscale;
    int heads, hd_log2, Tp;
    unsigned qkv_bytes;                // size of each of q / k / vt in bytes (buffer bound)
};

#define VB_VMW(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// wait until at most n vector-memory operations of this wave are outstanding (n wave-uniform, rounded down to a multiple of 3)
template <int N> __device__ __forceinline__ void vb_wait_const() {
    static_assert(N >= 0 && N % 3 == 0 && N <= 24, "");
    if constexpr (N == 0) VB_VMW(0); else if constexpr (N == 3) VB_VMW(3); else if constexpr (N == 6) VB_VMW(6); else if constexpr (N == 9) VB_VMW(9);
    else if constexpr (N == 12) VB_VMW(12); else if constexpr (N == 15) VB_VMW(15); else if constexpr (N == 18) VB_VMW(18);
    else if constexpr (N == 21) VB_VMW(21); else VB_VMW(24);
}
// STEADY: the count of the hidden loop's steady state ((NSLOT - 4) pieces x DPW in flight behind the step's two) - ONE compare in front of the
// switch, which hipcc turns into a tree of ~15 scalar branches (round 6: the tree ran at every boundary of the 5-slot ring, whose steady
// count was not the 12 tested here before)
template <int STEADY>
__device__ __forceinline__ void vb_wait_le(int n) {
    if (n == STEADY) { vb_wait_const<STEADY>(); return; }
    if (n >= 63) { VB_VMW(63); return; }
    switch (n / 3) {
        case 0: VB_VMW(0); break;   case 1: VB_VMW(3); break;   case 2: VB_VMW(6); break;   case 3: VB_VMW(9); break;
        case 4: VB_VMW(12); break;  case 5: VB_VMW(15); break;  case 6: VB_VMW(18); break;  case 7: VB_VMW(21); break;
        case 8: VB_VMW(24); break;  case 9: VB_VMW(27); break;  case 10: VB_VMW(30); break; case 11: VB_VMW(33); break;
        case 12: VB_VMW(36); break; case 13: VB_VMW(39); break; case 14: VB_VMW(42); break; case 15: VB_VMW(45); break;
        case 16: VB_VMW(48); break; case 17: VB_VMW(51); break; case 18: VB_VMW(54); break; case 19: VB_VMW(57); break;
        default: VB_VMW(60); break;
    }
}

#define VB_GELU_C0 (-2.3087653f)
#define VB_GELU_C1 (-0.10012561f)
// x * sigmoid form with a two-term exponent, coefficients fitted (minimax on [-10, 10]) to the exact erf GELU: max |error| 2.7e-4
__device__ __forceinline__ float vb_gelu16(float x) {
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * fmaf(x * x, VB_GELU_C1, VB_GELU_C0)));
}
// GELU for 16-bit storage (common.h:gelu_fast16) split into three stages of 3 VALU-class instructions per value, so that the
// hidden loop can hand them out between MFMAs: s0 -> (x2, p), s1 -> e = exp2(x * (p x2 + c0)), s2 -> x * rcp(1 + e).
__device__ __forceinline__ void gelu_s0(float x, float& x2, float& p) {
    x2 = fminf(x * x, 36.f);
    p = fmaf(x2, 0.0010142630555f, -0.1067757240036f);
}
__device__ __forceinline__ float gelu_s1(float x, float x2, float p) {
    return __builtin_amdgcn_exp2f(x * fmaf(p, x2, -2.3011213394584f));
}
__device__ __forceinline__ float gelu_s2(float x, float e) { return x * __builtin_amdgcn_rcpf(1.f + e); }

// ---- GELU on packed f16 pairs (round 5; the f16 default, LWDETR_VB_GELU16=0 turns it off). The same two-term form
// x * rcp(1 + exp2(x (c1 x^2 + c0))) evaluated in f16 on the value pairs that the fc2 operand needs anyway: per 8 values 4 v_cvt_pk +
// 4 v_pk_mul + 4 v_pk_fma + 4 v_pk_mul + 8 v_exp_f16 + 4 v_pk_add + 8 v_rcp_f16 + 4 v_pk_mul = 40 VALU-class instructions instead of
// 60 (the transcendentals have no packed form: SDWA word selects, in place). Inline asm throughout: the layer ticks must stay where
// they are put, no operand may carry an op_sel (tools/check_isa.py guards the packed-f32 forms; these packed-f16 ones are written
// straight so that the question never arises), and the constants must stay in registers (a compiler-materialised constant is a
// v_mov per use). f16 arithmetic: rms error of the GELU output 4.2e-4 instead of 2.7e-4 (f32 arithmetic, f16 result) on N(0, 1.5)
// inputs, maximum 2.7e-3 instead of 1.2e-3 - tests/vitblock_sim.py::gelu_vb16_packed is the bit-level model.
#define VB_G16_C0 0xc09ec09eu       // -2.30859375 (f16) twice
#define VB_G16_C1 0xae68ae68u       // -0.10009765625
#define VB_G16_ONE 0x3c003c00u
__device__ __forceinline__ unsigned g16_mul(unsigned a, unsigned b) { unsigned r; asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned g16_fma(unsigned a, unsigned c1s, unsigned c0v) {
    unsigned r; asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(c1s), "v"(c0v)); return r;
}
__device__ __forceinline__ unsigned g16_add(unsigned a, unsigned ones) { unsigned r; asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(ones)); return r; }
// the transcendentals of a group's four pairs: low halves first, then high halves - an SDWA write of half a register must not be
// followed directly by a read of that register (gfx940-family dst_sel forwarding hazard: one wait state; the hazard recognizer does
// not look inside inline asm), so each register's two instructions sit three instructions apart, and the next tick reads
// the registers in the same order; the block ENDS on a wait state, because what follows it is the compiler's choice
#define VB_G16_TRANS4(OP)                                                                                             \
    asm volatile(OP " %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t"                          \
                 OP " %1, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t"                          \
                 OP " %2, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t"                          \
                 OP " %3, %3 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t"                          \
                 OP " %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                          \
                 OP " %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                          \
                 OP " %2, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                          \
                 OP " %3, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"                          \
                 "s_nop 0"     /* the instruction hipcc places behind the statement may read %3 (advisor r5) */         \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
__device__ __forceinline__ void g16_exp2x4(unsigned& a, unsigned& b, unsigned& c, unsigned& d) { VB_G16_TRANS4("v_exp_f16_sdwa"); }
__device__ __forceinline__ void g16_rcpx4(unsigned& a, unsigned& b, unsigned& c, unsigned& d) { VB_G16_TRANS4("v_rcp_f16_sdwa"); }

template <typename T, int C, int NH, bool QKV, int WPC = 1, bool G16 = false>
__global__ __launch_bounds__(256, WPC) void vitblock_kernel(const VbParams p) {
    static_assert(!G16 || sizeof(T) == 2, "");
    typedef typename Vec<T>::v8 V8;
    static_assert(sizeof(T) == 2, "16-bit types only");
    constexpr int KS = C / 16;                  // k-steps of a K = C contraction = fragments per piece
    constexpr int NTI = C / 32;                 // 32-row tiles along C
    constexpr int NCH = C / 8;                  // hidden chunks of 32 units (4C / 32)
    constexpr int PIECE_B = KS * 1024;          // bytes per piece
    constexpr int DPW = KS / 4;                 // DMA wave-instructions per piece and wave
    constexpr int NSLOT = WPC == 2 ? 5 : (C == 192 ? 8 : 5);     // ring depth in pieces (WPC = 2: two workgroups per CU, 72 KB each)
    constexpr int VEC_F = 13 * C;
    constexpr int VEC_B = (VEC_F * 4 + 4095) / 4096 * 4096, VEC_DPW = VEC_B / 4096;
    constexpr int NP_PROJ = NTI, NP_HID = 2 * NCH, NP_QKV = QKV ? 3 * NTI : 0, NP = NP_PROJ + NP_HID + NP_QKV;
    constexpr int H0 = NP_PROJ, Q0 = NP_PROJ + NP_HID;
    constexpr int RD = NH == 2 ? 4 : 8;         // fragment read-ahead: an LDS round trip is ~170 cycles under load (see vitblock8_kernel)
    static_assert(DPW % 3 == 0 && VEC_DPW >= 1, "wait counts are kept in multiples of 3");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float* vec = (const float*)(smem + NSLOT * PIECE_B);
    const float* b1s = vec; const float* bps = vec + 4 * C; const float* g1s = bps + C; const float* rg1s = g1s + C;
    const float* b2s = rg1s + C; const float* rg2s = b2s + C; const float* g2s = rg2s + C; const float* bqs = g2s + C;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;

    // ---- this wave's tokens: 8-token units dealt evenly over all waves of the grid (at most 32 * NH per wave: host)
    const long U = p.M >> 3, nwv = (long)gridDim.x * 4, wg = (long)blockIdx.x * 4 + wave;
    const long u0 = wg * U / nwv, u1 = (wg + 1) * U / nwv;
    const long t0 = u0 * 8;
    const int nvalid = (int)(u1 - u0) * 8;

    // ---- weight stream: linear LDS-DMA, 1 KB per wave-instruction (inline asm: hipcc must not turn the pending pieces into
    // lgkmcnt(0) drains of the fragment reads, cf. mlp.hip). Piece i lives in ring slot i % NSLOT.
    const char* wsrc = (const char*)p.wstream;
    auto dma1k = [&](const char* src_uniform, unsigned voff, unsigned lds_dst) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)src_uniform);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)src_uniform >> 32));
        const char* sp = (const char*)(((uintptr_t)hi << 32) | lo);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(sp) : "memory");
    };
    // the next piece to issue: its number, its source and its ring slot are carried along (no multiplication / modulo per piece in the run-time loop)
    int issued = 0;
    const char* isrc = wsrc;
    unsigned ioff = 0;                          // byte offset of the slot of piece `issued`
    auto dma_next = [&]() {
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const unsigned kb = (unsigned)(wave * DPW + i) * 1024u;
            dma1k(isrc, kb + lane16, lds0 + ioff + kb);
        }
        ++issued; isrc += PIECE_B; ioff += PIECE_B;
        if (ioff == NSLOT * PIECE_B) ioff = 0;
    };
    auto issued_is = [&](int n) { issued = n; isrc = wsrc + (size_t)n * PIECE_B; ioff = (unsigned)(n % NSLOT) * PIECE_B; };      // (re-stated as constants)
    // step boundary: the step consumes pieces [a, b), everything below a is dead. `extra` = vector-memory operations this wave
    // is KNOWN to have issued after its DMA of piece b - 1 besides later pieces (a lower bound is safe, it only waits longer).
#ifdef LWDETR_VB_TIMING
    unsigned long long tt_wait = 0, tt_bar = 0;
#endif
    auto boundary = [&](int a, int b, int extra) {
        // nothing of the step before may be scheduled behind this point: the MFMAs that consume the fragments of the pieces below
        // `a` - and with them the lgkmcnt wait for those reads - stay in front of the barrier that frees their ring slots (hipcc
        // does not see the DMA; gemm.hip's convolution kernel shows what happens otherwise)
        __builtin_amdgcn_sched_barrier(0);
#ifdef LWDETR_VB_TIMING
        const unsigned long long ta = __builtin_amdgcn_s_memrealtime();
#endif
        vb_wait_le<(NSLOT - 4) * DPW>((issued - b) * DPW + extra);
#ifdef LWDETR_VB_TIMING
        const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
#endif
        // Every fragment read of the dead pieces has RETURNED before this wave signals: the barrier releases their ring slots to the DMA issued
        // right behind it. The sched_barrier above does not guarantee that - instruction selection may still sink the MFMAs, and with them the
        // lgkmcnt waits of their operands, below this point (round 6: a build in which 11 of the pre-step's 12 MFMAs sat behind the next
        // boundary, their reads issued but not waited for, failed parity once in a few runs with two workgroups per CU). Free when
        // nothing is outstanding, which is the designed state.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef LWDETR_VB_TIMING
        tt_wait += tb - ta; tt_bar += __builtin_amdgcn_s_memrealtime() - tb;
#endif
        int lim = a + NSLOT; lim = lim < NP ? lim : NP;
#if LWDETR_VB_ABLATE & 4
        if (a >= H0) { issued = lim; return; }
#endif
        while (issued < lim) dma_next();
    };
    auto frag = [&](int piece, int f) -> V8 {
        return *(const V8*)(smem + ((unsigned)piece % NSLOT) * PIECE_B + f * 1024 + lane16);
    };
    auto frag_o = [&](unsigned off, int f) -> V8 { return *(const V8*)(smem + off + f * 1024 + lane16); };      // by slot byte offset
    auto next_off = [](unsigned off) { off += PIECE_B; return off == NSLOT * PIECE_B ? 0u : off; };

    VB_TS(0);
    // ---- prologue: vectors + the first NSLOT pieces in flight, then the attention rows as B fragments
    {
        const char* vsrc = (const char*)p.vec;
#pragma unroll
        for (int i = 0; i < VEC_DPW; ++i) {
            const unsigned kb = (unsigned)(wave * VEC_DPW + i) * 1024u;
            dma1k(vsrc, kb + lane16, lds0 + NSLOT * PIECE_B + kb);
        }
        while (issued < NSLOT) dma_next();
    }
    const T* att_w = (const T*)p.att + t0 * p.ldatt;
    T* x_w = (T*)p.x + t0 * p.ldx;
    const __amdgpu_buffer_rsrc_t r_att = __builtin_amdgcn_make_buffer_rsrc((void*)att_w, 0, (int)(nvalid * p.ldatt * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)x_w, 0, (int)(nvalid * p.ldx * 2), 0x00020000);
    V8 xf[NH][KS];                              // attention rows, later LN(x1), later LN'(x): B operands, k-run of 8 per k-step
#pragma unroll
    for (int th = 0; th < NH; ++th)
#pragma unroll
        for (int t = 0; t < KS; ++t)
            xf[th][t] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(
                r_att, (unsigned)(((32 * th + j) * p.ldatt + 16 * t + 8 * h) * 2), 0, 0));

    f32x16 acc2[NTI][NH];
    // ---- attention output projection: D[channel][token], Wp pieces of 32 output channels, 2 pieces per step. The accumulators
    // start at x / gamma1 + bp (x in accumulator layout: 4 channels 32 n + 8 b + 4 h .. of token j), so x1 = gamma1 * acc and the
    // x rows are consumed before the first MFMA (no second copy of the residual stream in registers).
    {
        u32x4 xv[NH][NTI][2];                   // 16-byte loads: channels 32 n + 16 jb + 8 h .. + 7 of token j
#pragma unroll
        for (int th = 0; th < NH; ++th)
#pragma unroll
            for (int n = 0; n < NTI; ++n)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
                    xv[th][n][jb] = __builtin_amdgcn_raw_buffer_load_b128(
                        r_x, (unsigned)(((32 * th + j) * p.ldx + 32 * n + 16 * jb + 8 * h) * 2), 0, 0);
        // loads issued after the DMA of the initial pieces: the attention and x rows
        boundary(0, 2, NH * KS + NH * NTI * 2);
        VB_TS(1);
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            f32x16 na[NH];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int th = 0; th < NH; ++th) {
                    // the half-wave exchange of vb_rows8 is its own inverse: back to this lane's accumulator rows
                    const u32x4 own = vb_rows8(xv[th][n][jb][0], xv[th][n][jb][1], xv[th][n][jb][2], xv[th][n][jb][3]);
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int b = 2 * jb + bb, c0 = 32 * n + 8 * b + 4 * h;
                        const f32x4 bbv = *(const f32x4*)(bps + c0), rg = *(const f32x4*)(rg1s + c0);
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            // (a scalar copy first: __builtin_bit_cast of an ext-vector ELEMENT reads element 0 whatever the index - hipcc 7.2)
                            const unsigned ow_ = own[2 * bb + d];
                            const typename Pk<T>::v2 v2 = __builtin_bit_cast(typename Pk<T>::v2, ow_);
                            na[th][4 * b + 2 * d] = fmaf(to_f32<T>(v2[0]), rg[2 * d], bbv[2 * d]);
                            na[th][4 * b + 2 * d + 1] = fmaf(to_f32<T>(v2[1]), rg[2 * d + 1], bbv[2 * d + 1]);
                        }
                    }
                }
#pragma unroll
            for (int th = 0; th < NH; ++th) {
                asm volatile("" : "+a"(na[th]));      // complete tile, in the accumulator file (see the LayerNorm section)
                acc2[n][th] = na[th];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    VB_TS(2);
#pragma unroll
    for (int s = 0; s < NTI / 2; ++s) {
        if (s > 0) boundary(2 * s, 2 * s + 2, 2 * s + 1 < NSLOT ? NH * KS + NH * NTI * 2 : 0);
        {   // 2 KS fragments through the read-ahead ring, one read per fragment retired (hipcc would hoist all 2 KS reads)
            V8 fr[RD];
#pragma unroll
            for (int i = 0; i < RD; ++i) fr[i] = frag(2 * s + i / KS, i % KS);
#pragma unroll
            for (int fi = 0; fi < 2 * KS; ++fi) {
                const int n = 2 * s + fi / KS, t = fi % KS;
                const V8 a = fr[fi % RD];
#pragma unroll
                for (int th = 0; th < NH; ++th) acc2[n][th] = Mma32<T>::k16(a, xf[th][t], acc2[n][th]);
                if (fi + RD < 2 * KS) fr[fi % RD] = frag(2 * s + (fi + RD) / KS, (fi + RD) % KS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    VB_TS(3);
    // ---- x1 = gamma1 * acc rounded to the storage type; LayerNorm (affine folded into W1 / b1) -> B fragments; the fc2
    // accumulators start at x1 / gamma2 + b2 (the epilogue is out = gamma2 * acc). x1 is held as packed 16-bit pairs between
    // the three passes (sum, variance, normalise): 48 registers per token half instead of 96 f32 values.
    auto unpack2 = [](unsigned w, float& a, float& b) {
        const typename Pk<T>::v2 v = __builtin_bit_cast(typename Pk<T>::v2, w);
        a = to_f32<T>(v[0]); b = to_f32<T>(v[1]);
    };
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        unsigned xp[NTI][8];                    // dword d of tile n: registers 2 d, 2 d + 1 of the accumulator tile
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 gg = *(const f32x4*)(g1s + 32 * n + 8 * b + 4 * h);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned w = pack2<T>(gg[2 * d] * acc2[n][th][4 * b + 2 * d], gg[2 * d + 1] * acc2[n][th][4 * b + 2 * d + 1]);
                    xp[n][2 * b + d] = w;
                    float v0, v1; unpack2(w, v0, v1);
                    s += v0 + v1;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        s += __shfl_xor(s, 32);
        const float mean = s * (1.f / C);
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n)
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                float v0, v1; unpack2(xp[n][d], v0, v1);
                v0 -= mean; v1 -= mean;
                v = fmaf(v0, v0, v); v = fmaf(v1, v1, v);
            }
        __builtin_amdgcn_sched_barrier(0);
        v += __shfl_xor(v, 32);
        const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps), nmr = -mean * rstd;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            u32x4 w0, w1;
            f32x16 na;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c0 = 32 * n + 8 * b + 4 * h;
                const f32x4 rg = *(const f32x4*)(rg2s + c0), b2 = *(const f32x4*)(b2s + c0);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    float v0, v1; unpack2(xp[n][2 * b + d], v0, v1);
                    na[4 * b + 2 * d] = fmaf(v0, rg[2 * d], b2[2 * d]);
                    na[4 * b + 2 * d + 1] = fmaf(v1, rg[2 * d + 1], b2[2 * d + 1]);
                    const unsigned nw = pack2<T>(fmaf(v0, rstd, nmr), fmaf(v1, rstd, nmr));
                    if (b < 2) w0[2 * b + d] = nw; else w1[2 * (b - 2) + d] = nw;
                }
            }
            asm volatile("" : "+a"(na));            // the new accumulator tile is complete (and in the accumulator file) here
            acc2[n][th] = na;
            xf[th][2 * n] = __builtin_bit_cast(V8, w0);
            xf[th][2 * n + 1] = __builtin_bit_cast(V8, w1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    VB_TS(4);
    // ---- hidden loop, software pipelined. Pieces after the projection: W1c(0), W1c(1), then (W2c(k-1), W1c(k+1)) for k = 1 ..
    // NCH-2, then W2c(NCH-2), W2c(NCH-1). Iteration k: GELU(k) on the VALU, fc2(k-1) and fc1(k+1) on the matrix pipe.
    f32x16 acc1[2][NH];
    unsigned g16_c0 = 0, g16_c1 = 0, g16_one = 0;      // G16: the packed-f16 GELU's constants, produced by asm so that they STAY in registers
    if constexpr (G16) {
        asm volatile("v_mov_b32 %0, 0xc09ec09e" : "=v"(g16_c0));
        asm volatile("s_mov_b32 %0, 0xae68ae68" : "=s"(g16_c1));
        asm volatile("s_mov_b32 %0, 0x3c003c00" : "=s"(g16_one));
    }
    u32x4 hf[2][NH][2];                         // GELU output as B operands (packed pairs): [buffer][token half][k-step of the chunk]
    auto bias16 = [&](const float* src) -> f32x16 {      // src[8 b + 4 h + e] -> register 4 b + e
        f32x16 r;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 v = *(const f32x4*)(src + 8 * b + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[4 * b + e] = v[e];
        }
        return r;
    };
    // one pipelined iteration. CUR: acc1 / hf buffer of chunk k; o2 / o1: ring slots (byte offsets) of the pieces W2c(k-1) / W1c(k+1) (ignored when the half is off)
    auto iter = [&](auto cur_tag, auto fc2_tag, auto fc1_tag, unsigned o2, unsigned o1, int k) {      // o2 / o1: slot offsets of the pieces W2c(k-1) / W1c(k+1)
        constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
        constexpr bool DO2 = decltype(fc2_tag)::value, DO1 = decltype(fc1_tag)::value;
        constexpr int NF2 = DO2 ? 2 * NTI : 0, NF1 = DO1 ? KS : 0, NF = NF2 + NF1, S = NF * NH;
        // GELU of chunk k in "layer ticks": a tick applies ONE instruction of the GELU chain to the 8 values of a group (one
        // (token half, k-step) B operand of fc2), so consecutive instructions of a chain sit a whole MFMA slot apart - with one
        // wave per SIMD nothing else hides VALU / transcendental result latency (3-stage ticks on 2 values measured 1.8 us per
        // iteration). 8 layers x 2 NH groups of ticks over the iteration's MFMA slots.
        constexpr int NG = 2 * NH, NL = 8, TK = NL * NG;
        auto fragi = [&](int i) -> V8 { return i < NF2 ? frag_o(o2, i) : frag_o(o1, i - NF2); };
        f32x16 bias = {};
        V8 fr[RD];
#pragma unroll
        for (int i = 0; i < RD; ++i) if (i < NF) fr[i] = fragi(i);
        float ga[8], gb[8];
        unsigned gx[4], gq[4];                 // G16: the group's value pairs / the running term
#define VB_PIN8(v) asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
        // GELU for 16-bit storage in its two-term form x * sigmoid(-x (c0 + c1 x^2)) (vb_gelu16, max |error| 2.7e-4 - a tenth of
        // what 16-bit arithmetic costs end to end): 7 instructions per value instead of 9
        auto tick = [&](auto ti_tag) {
            constexpr int ti = decltype(ti_tag)::value, grp = ti / NL, L = ti % NL, th = grp >> 1, r0 = 8 * (grp & 1);
            if constexpr (G16) {
                // packed-f16 form: gx = the four value pairs of the group (what fc2 multiplies is their GELU), gq = the running term
                if constexpr (L == 4) g16_exp2x4(gq[0], gq[1], gq[2], gq[3]);
                else if constexpr (L == 6) g16_rcpx4(gq[0], gq[1], gq[2], gq[3]);
                else {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        if constexpr (L == 0) gx[d] = pack2<T>(acc1[CUR][th][r0 + 2 * d], acc1[CUR][th][r0 + 2 * d + 1]);
                        else if constexpr (L == 1) gq[d] = g16_mul(gx[d], gx[d]);
                        else if constexpr (L == 2) gq[d] = g16_fma(gq[d], g16_c1, g16_c0);
                        else if constexpr (L == 3) gq[d] = g16_mul(gx[d], gq[d]);
                        else if constexpr (L == 5) gq[d] = g16_add(gq[d], g16_one);
                        else gq[d] = g16_mul(gx[d], gq[d]);
                    }
                }
                if constexpr (L == 0) asm volatile("" : "+v"(gx[0]), "+v"(gx[1]), "+v"(gx[2]), "+v"(gx[3]));
                if constexpr (L == 7) hf[CUR][th][grp & 1] = u32x4{gq[0], gq[1], gq[2], gq[3]};
            } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float x = acc1[CUR][th][r0 + u];
                if constexpr (L == 0) ga[u] = x * x;
                else if constexpr (L == 1) gb[u] = fmaf(ga[u], VB_GELU_C1, VB_GELU_C0);
                else if constexpr (L == 2) gb[u] = x * gb[u];
                else if constexpr (L == 3) gb[u] = __builtin_amdgcn_exp2f(gb[u]);
                else if constexpr (L == 4) gb[u] = 1.f + gb[u];
                else if constexpr (L == 5) gb[u] = __builtin_amdgcn_rcpf(gb[u]);
                else if constexpr (L == 6) gb[u] = x * gb[u];
            }
            if constexpr (L < 1) VB_PIN8(ga);
            else if constexpr (L < 7) VB_PIN8(gb);
            else {
                u32x4 w;
#pragma unroll
                for (int d = 0; d < 4; ++d) w[d] = pack2<T>(gb[2 * d], gb[2 * d + 1]);
                asm volatile("" : "+v"(w));
                hf[CUR][th][grp & 1] = w;
            }
            }
        };
        static_assert((TK + S - 1) / S <= 2, "at most two GELU ticks per MFMA slot");
        vb_static_for<S>([&](auto m_tag) {
            constexpr int m = decltype(m_tag)::value;
            constexpr int t_lo = m * TK / S, t_hi = (m + 1) * TK / S;
#if !(LWDETR_VB_ABLATE & 1)
            if constexpr (t_lo < t_hi) tick(std::integral_constant<int, t_lo>{});
            if constexpr (t_lo + 1 < t_hi) tick(std::integral_constant<int, t_lo + 1>{});
#endif
            constexpr int fi = m / NH, th = m % NH;
            if constexpr (DO1 && m == NF2 * NH - (NF2 ? 4 : 0)) bias = bias16(b1s + (k + 1) * 32);     // short live range: just ahead of fc1
            const V8 a = fr[fi % RD];
#if LWDETR_VB_ABLATE & 2
            if constexpr (fi < NF2) { asm volatile("" : "+a"(acc2[fi % NTI][th]) : "v"(a), "v"(hf[NXT][th][fi / NTI])); }
            else { if constexpr (fi == NF2) acc1[NXT][th] = bias; asm volatile("" : "+v"(acc1[NXT][th]) : "v"(a), "v"(xf[th][fi - NF2])); }
#else
            if constexpr (fi < NF2) {
                constexpr int kap = fi / NTI, n = fi % NTI;
                acc2[n][th] = Mma32<T>::k16(a, __builtin_bit_cast(V8, hf[NXT][th][kap]), acc2[n][th]);
            } else {
                constexpr int t = fi - NF2;
                if constexpr (t == 0) acc1[NXT][th] = Mma32<T>::k16(a, xf[th][t], bias);
                else acc1[NXT][th] = Mma32<T>::k16(a, xf[th][t], acc1[NXT][th]);
            }
#endif
#if !(LWDETR_VB_ABLATE & 8)
            if constexpr (th == NH - 1 && fi + RD < NF) fr[fi % RD] = fragi(fi + RD);
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<bool, true> Yes;
    typedef std::integral_constant<bool, false> No;
    { constexpr int il = NTI - 2 + NSLOT; issued_is(il < NP ? il : NP); }      // (what it is after the projection's last boundary, as a constant)
    {   // pre-step: fc1(0)
        boundary(H0, H0 + 1, 0);
        const f32x16 bias = bias16(b1s);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const V8 a = frag(H0, t);
#pragma unroll
            for (int th = 0; th < NH; ++th) acc1[0][th] = Mma32<T>::k16(a, xf[th][t], t == 0 ? bias : acc1[0][th]);
        }
    }
    VB_TS(5);
    boundary(H0 + 1, H0 + 2, 0);
    iter(I0{}, No{}, Yes{}, 0u, (unsigned)((H0 + 1) % NSLOT) * PIECE_B, 0);
    VB_TS(6);
#ifdef LWDETR_VB_TIMING
    tt_wait = 0; tt_bar = 0;
#endif
    unsigned co = (unsigned)((H0 + 2) % NSLOT) * PIECE_B;      // slot of the next piece to consume (piece H0 + 2 k at the head of iteration k)
#pragma unroll 1
    for (int k = 1; k < NCH - 1; k += 2) {
        boundary(H0 + 2 * k, H0 + 2 * k + 2, 0);
        { const unsigned o1 = next_off(co); iter(I1{}, Yes{}, Yes{}, co, o1, k); co = next_off(o1); }
        boundary(H0 + 2 * k + 2, H0 + 2 * k + 4, 0);
        { const unsigned o1 = next_off(co); iter(I0{}, Yes{}, Yes{}, co, o1, k + 1); co = next_off(o1); }
    }
    VB_TS(7);
#ifdef LWDETR_VB_TIMING
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane == 0) { g_vb_timing[blockIdx.x != 0][wave][13] = tt_wait; g_vb_timing[blockIdx.x != 0][wave][14] = tt_bar; }
#endif
    // the loop's last boundary was (H0 + 2 NCH - 4, ..): `issued` is a compile-time constant again from here on (hipcc does not see it through the
    // run-time loop), and every later boundary - wait count, ring slots, pieces to issue - folds
    { constexpr int il = H0 + 2 * NCH - 4 + NSLOT; issued_is(il < NP ? il : NP); }
    boundary(H0 + 2 * NCH - 2, H0 + 2 * NCH - 1, 0);
    iter(I1{}, Yes{}, No{}, (unsigned)((H0 + 2 * NCH - 2) % NSLOT) * PIECE_B, 0u, NCH - 1);
    VB_TS(8);
    {   // post-step: fc2(NCH - 1)
        boundary(H0 + 2 * NCH - 1, H0 + 2 * NCH, 0);
#pragma unroll
        for (int fi = 0; fi < 2 * NTI; ++fi) {
            const V8 a = frag(H0 + 2 * NCH - 1, fi);
            const int kap = fi / NTI, n = fi % NTI;
#pragma unroll
            for (int th = 0; th < NH; ++th) acc2[n][th] = Mma32<T>::k16(a, __builtin_bit_cast(V8, hf[1][th][kap]), acc2[n][th]);
        }
    }

    VB_TS(9);
    // ---- epilogue: out = gamma2 * acc (rounded), stores, statistics of the new rows, LN'(x) as the next B operand
    const __amdgpu_buffer_rsrc_t r_o2 = __builtin_amdgcn_make_buffer_rsrc(
        p.out2 ? (void*)((T*)p.out2 + t0 * p.ld2) : (void*)x_w, 0, p.out2 ? (int)(nvalid * p.ld2 * 2) : 0, 0x00020000);
    const bool has_o2 = p.out2 != nullptr;
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                unsigned pw[4];                     // own packed pairs: registers 8 jb + 2 d, + 1
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int b = 2 * jb + bb;
                    const f32x4 gg = *(const f32x4*)(g2s + 32 * n + 8 * b + 4 * h);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const unsigned w = pack2<T>(gg[2 * d] * acc2[n][th][4 * b + 2 * d], gg[2 * d + 1] * acc2[n][th][4 * b + 2 * d + 1]);
                        pw[2 * bb + d] = w;
                        const typename Pk<T>::v2 v2 = __builtin_bit_cast(typename Pk<T>::v2, w);
                        const float v0 = to_f32<T>(v2[0]), v1 = to_f32<T>(v2[1]);
                        acc2[n][th][4 * b + 2 * d] = v0; acc2[n][th][4 * b + 2 * d + 1] = v1;
                        s += v0 + v1;
                    }
                }
                const u32x4 ow = vb_rows8(pw[0], pw[1], pw[2], pw[3]);      // channels 32 n + 16 jb + 8 h .. + 7 of token j
                const int c0 = 32 * n + 16 * jb + 8 * h;
                __builtin_amdgcn_raw_buffer_store_b128(ow, r_x, (unsigned)(((32 * th + j) * p.ldx + c0) * 2), 0, 0);
                if (has_o2) __builtin_amdgcn_raw_buffer_store_b128(ow, r_o2, (unsigned)(((32 * th + j) * p.ld2 + c0) * 2), 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.stats_out || QKV) {
            s += __shfl_xor(s, 32);
            const float mean = s * (1.f / C);
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < NTI; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float dl = acc2[n][th][e] - mean; v += dl * dl; }
            v += __shfl_xor(v, 32);
            const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps_next), nmr = -mean * rstd;
            if (p.stats_out && h == 0 && 32 * th + j < nvalid) {
                float* so = p.stats_out + 2 * (t0 + 32 * th + j);
                so[0] = mean; so[1] = rstd;
            }
            if (QKV) {
#pragma unroll
                for (int n = 0; n < NTI; ++n)
#pragma unroll
                    for (int be = 0; be < 2; ++be)
                        {
                            u32x4 w;
#pragma unroll
                            for (int d = 0; d < 4; ++d) {
                                const int r = (2 * be + (d >> 1)) * 4 + (d & 1) * 2;
                                w[d] = pack2<T>(fmaf(acc2[n][th][r], rstd, nmr), fmaf(acc2[n][th][r + 1], rstd, nmr));
                            }
                            xf[th][2 * n + be] = __builtin_bit_cast(V8, w);
                        }
            }
        }
    }

    VB_TS(10);
    if (QKV) {
        // ---- chained norm1 + QKV of the next block: pieces of 32 features (q: 0 .. NTI-1, k: NTI .. 2 NTI-1, v: 2 NTI ..).
        // Q, K: D[feature][token] -> (B, heads, Tp, hd); V: operands swapped, D[token][feature] -> V^T (B, heads, hd, Tp).
        const __amdgpu_buffer_rsrc_t r_q = __builtin_amdgcn_make_buffer_rsrc(p.q, 0, (int)p.qkv_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_k = __builtin_amdgcn_make_buffer_rsrc(p.k, 0, (int)p.qkv_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc(p.vt, 0, (int)p.qkv_bytes, 0x00020000);
        const int hd = 1 << p.hd_log2;
        // Store addresses: a per-lane byte offset computed ONCE (voffset; 0x80000000 = no token: out of the buffer's range) + a wave-uniform byte
        // offset per group (soffset: SALU only) - no vector address arithmetic between the MFMAs. q / k (B, heads, Tp, hd): the lane's 8 features
        // f = K + 8 h with K a multiple of 16 and hd >= 8, so column(f) = column(K) + column(8 h); v^T (B, heads, hd, Tp): row head * hd + d = f itself.
        auto qk_col = [&](int f) -> unsigned { return (unsigned)((((long)(f >> p.hd_log2) * p.Tp) << p.hd_log2) + (f & (hd - 1))); };
        unsigned vo_qk[NH];
#pragma unroll
        for (int th = 0; th < NH; ++th) {
            const unsigned tok = (unsigned)t0 + 32 * th + j;
            const unsigned img = tok / (unsigned)p.Tp, wi = tok - img * (unsigned)p.Tp;
            vo_qk[th] = 32 * th + j < nvalid ? ((unsigned)(((long)img * p.heads * p.Tp + wi) << p.hd_log2) + qk_col(8 * h)) * 2u : 0x80000000u;
        }
        constexpr int XST = NH * NTI * 2, SPS = 2 * NH * 2, LAG = (NSLOT - 2) / 2;
        const int est = XST * (has_o2 ? 2 : 1);
        unsigned vo_v[NH][2];                     // v^T: 8-token runs 32 th + 16 jb + 8 h .. of this lane, feature row j
#pragma unroll
        for (int th = 0; th < NH; ++th)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int tl = 32 * th + 16 * jb + 8 * h;
                const unsigned tk = (unsigned)t0 + tl;
                const unsigned im = tk / (unsigned)p.Tp, wv = tk - im * (unsigned)p.Tp;
                vo_v[th][jb] = tl < nvalid ? ((unsigned)((long)im * p.heads * hd * p.Tp + wv) + (unsigned)j * (unsigned)p.Tp) * 2u : 0x80000000u;
            }
        // (pinned: computed HERE, once - left alone, hipcc sinks the divisions into the steps that first use the offsets and reloads their inputs
        // from scratch there, and a scratch reload is a vmcnt(0) in the middle of the weight ring)
#pragma unroll
        for (int th = 0; th < NH; ++th) asm volatile("" : "+v"(vo_qk[th]), "+v"(vo_v[th][0]), "+v"(vo_v[th][1]));
        // A step = two pieces = 2 KS NH MFMAs. Round 6 (profiles/r6e_*): with the pieces one after the other - fragment round trip, KS dependent
        // MFMAs, scale / conversion / half-wave exchange / addresses / stores - a step took 1.33 us for 0.37 us of matrix time (one wave per SIMD:
        // nothing else fills the gaps), and the stores themselves cost nothing (ablated: the same). Now the step is software-pipelined like the
        // hidden loop: the finished accumulators of step s - 1 (ping-pong) are turned into stores in GROUPS (8 accumulator registers -> one 16-byte
        // store per lane) whose work is handed out in six LAYERS over the MFMA slots of step s - at most ~5 single-issue instructions fit
        // beside a 32x32x16 MFMA of a wave that has its SIMD to itself, and an accumulator read or a transcendental counts double
        // (tools/microbench/filler_bench.hip, profiles/r6e_*). The step's two pieces alternate (two independent accumulator chains) and start
        // from zero: the bias is added in the group (f32, before the scale), not loaded into 32 accumulator registers per step.
        constexpr int NS = NP_QKV / 2, NGR = 4 * NH, SLOTS = 2 * KS, GAP = SLOTS / NGR, NLAY = 6;
        constexpr int RDQ = 4;                    // fragment read-ahead of this phase in MFMA slots
        static_assert(NTI % 2 == 0 && SLOTS % NGR == 0 && GAP >= 3, "both pieces of a step belong to one of q / k / v");
        f32x16 qacc[2][2][NH];                    // [ping-pong][piece of the step][token half]
        float gv[8];                              // the group in flight: values, bias, packed pairs
        f32x4 gb0, gb1;
        float gbv = 0.f;
        // (the step index is a compile-time constant throughout - q / k / v, ping-pong buffer and column offsets fold - NS steps of straight code)
        auto layer = [&](auto sh_tag, auto g_tag, auto l_tag) {      // sh: the held step, g: its group, l: the layer
            constexpr int sh = decltype(sh_tag)::value, BUF = sh & 1, g = decltype(g_tag)::value, L = decltype(l_tag)::value;
            constexpr int pp = g / (2 * NH), jb = (g / NH) % 2, th = g % NH;
            constexpr int pi = 2 * sh + pp, sg = pi / NTI, nl0 = (pi - sg * NTI) * 32;
            // Instruction selection places what has no side effect wherever its operands allow - the conversions of a whole step ended up in
            // front of its first MFMA, behind 24 hoisted fragment reads that no longer fitted the registers - and sched_barrier only binds the
            // scheduler that runs after it. Every layer therefore BEGINS at an empty asm its inputs pass through (the accumulator tuple for the
            // two reading layers) and ends at one its results pass through.
            if constexpr (L <= 1) asm volatile("" : "+a"(qacc[BUF][pp][th]));
            else asm volatile("" : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7]));
            if constexpr (L == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[e] = qacc[BUF][pp][th][8 * jb + e];
                if constexpr (sg < 2) {           // rows 16 jb + 4 h + {0..3} and + 8 of the piece: features of D[feature][token]
                    gb0 = *(const f32x4*)(bqs + sg * C + nl0 + 16 * jb + 4 * h);
                    gb1 = *(const f32x4*)(bqs + sg * C + nl0 + 16 * jb + 8 + 4 * h);
                } else {
                    // (the lane index from the hardware: the compiler keeps `j` in scratch here, and a scratch reload is a vmcnt(0) - it cannot
                    // count the DMA it does not see - in the middle of the weight ring)
                    unsigned ln;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                    gbv = bqs[2 * C + nl0 + (int)(ln & 31u)];      // D[token][feature]: one column per lane
                }
            } else if constexpr (L == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[4 + e] = qacc[BUF][pp][th][8 * jb + 4 + e];
            } else if constexpr (L == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[e] += sg < 2 ? gb0[e] : gbv;
            } else if constexpr (L == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[4 + e] += sg < 2 ? gb1[e] : gbv;
            } else if constexpr (L == 4) {
                if constexpr (sg == 0) {            // Q is pre-scaled
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[e] *= p.qscale;
                }
            } else {
                const u32x4 ow = vb_rows8(pack2<T>(gv[0], gv[1]), pack2<T>(gv[2], gv[3]), pack2<T>(gv[4], gv[5]), pack2<T>(gv[6], gv[7]));
                if constexpr (sg < 2) {
                    const unsigned so = __builtin_amdgcn_readfirstlane(qk_col(nl0 + 16 * jb) * 2u);
                    if constexpr (sg == 0) __builtin_amdgcn_raw_buffer_store_b128(ow, r_q, vo_qk[th], so, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(ow, r_k, vo_qk[th], so, 0);
                } else {
                    const unsigned so = __builtin_amdgcn_readfirstlane((unsigned)nl0 * (unsigned)p.Tp * 2u);
                    __builtin_amdgcn_raw_buffer_store_b128(ow, r_v, vo_v[th][jb], so, 0);
                }
            }
            if constexpr (L < NLAY - 1)            // the layer stays in its slot
                asm volatile("" : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7]));
        };
        vb_static_for<NS>([&](auto s_tag) {
            constexpr int sidx = decltype(s_tag)::value, CUR = sidx & 1;
            constexpr bool HELD = sidx > 0;
            // vector-memory operations newer than the DMA of this step's last piece (issued at boundary s - LAG, or in the hidden loop) besides
            // later pieces: the stores handed out during the steps s - LAG .. s - 1 = those of steps s - LAG - 1 .. s - 2, and the epilogue's when
            // the piece went out before them (exact: every store is issued unconditionally, invalid lanes store out of range)
            constexpr int fl = sidx - 1 < LAG ? (sidx - 1 > 0 ? sidx - 1 : 0) : LAG;
            boundary(Q0 + 2 * sidx, Q0 + 2 * sidx + 2, (sidx < LAG ? est : 0) + SPS * fl);
            constexpr int piece = Q0 + 2 * sidx, sg = (2 * sidx) / NTI;
            auto fragq = [&](int i) -> V8 { return frag(piece + (i & 1), i >> 1); };      // slot i: piece i & 1, k-step i >> 1
            V8 fr[RDQ];
#pragma unroll
            for (int i = 0; i < RDQ; ++i) fr[i] = fragq(i);
            vb_static_for<SLOTS>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, pp = i & 1, t = i >> 1;
                // (the slot's MFMAs stay in the slot - between an empty asm their weight fragment passes through and one their result passes
                // through: left to itself, instruction selection ran the twelve MFMAs of one accumulator back to back behind 24 hoisted reads)
                asm volatile("" : "+v"(fr[i % RDQ]) :: "memory");
                const V8 a = fr[i % RDQ];
                // Q, K: D[feature][token]; V: operands swapped, D[token][feature]
#pragma unroll
                for (int th = 0; th < NH; ++th) {
                    const f32x16 zero = {};
#if LWDETR_VB_ABLATE & 2
                    if constexpr (t == 0) qacc[CUR][pp][th] = zero;
                    asm volatile("" : "+a"(qacc[CUR][pp][th]) : "v"(a), "v"(xf[th][t]));
#else
                    if constexpr (sg < 2) qacc[CUR][pp][th] = Mma32<T>::k16(a, xf[th][t], t == 0 ? zero : qacc[CUR][pp][th]);
                    else qacc[CUR][pp][th] = Mma32<T>::k16(xf[th][t], a, t == 0 ? zero : qacc[CUR][pp][th]);
                    asm volatile("" : "+a"(qacc[CUR][pp][th]));
#endif
                }
#if !(LWDETR_VB_ABLATE & 8)
                if constexpr (i + RDQ < SLOTS) fr[i % RDQ] = fragq(i + RDQ);
#endif
                if constexpr (HELD) {
                    constexpr int g = i / GAP, o = i % GAP;
                    vb_static_for<NLAY>([&](auto l_tag) {
                        if constexpr (decltype(l_tag)::value * GAP / NLAY == o) layer(std::integral_constant<int, sidx - 1>{}, std::integral_constant<int, g>{}, l_tag);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        vb_static_for<NGR>([&](auto g_tag) {
            vb_static_for<NLAY>([&](auto l_tag) { layer(std::integral_constant<int, NS - 1>{}, g_tag, l_tag); });
        });
    }
    VB_TS(11);
}

// ---------------------------------------------------------------------------------------------------- norm1 + QKV alone
// Block 0 of the ViT has no block kernel in front of it: its norm1 + QKV (vit.py:199, :123-130) ran as a LayerNorm launch + a GEMM
// (12 + 59 us at BASELINE config 2). This is the chained QKV phase of vitblock_kernel on its own: the wave's token rows come from
// memory as B fragments (natural k order: the weights are packed in natural order too, kernels.py:pack_vit_qkv), LayerNorm in
// registers (affine folded into the weights), Q / K as (B, heads, Tp, hd), V^T as (B, heads, hd, Tp).
template <typename T, int C, int NH>
__global__ __launch_bounds__(256, 1) void vit_qkv_kernel(const VbParams p) {
    typedef typename Vec<T>::v8 V8;
    static_assert(sizeof(T) == 2, "16-bit types only");
    constexpr int KS = C / 16, NTI = C / 32;
    constexpr int PIECE_B = KS * 1024, DPW = KS / 4;
    constexpr int NSLOT = C == 192 ? 8 : 5;
    constexpr int VEC_B = (3 * C * 4 + 4095) / 4096 * 4096, VEC_DPW = VEC_B / 4096;
    constexpr int NP = 3 * NTI;
    constexpr int RD = NH == 2 ? 4 : 8;
    static_assert(DPW % 3 == 0, "wait counts are kept in multiples of 3");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float* bqs = (const float*)(smem + NSLOT * PIECE_B);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const long U = p.M >> 3, nwv = (long)gridDim.x * 4, wg = (long)blockIdx.x * 4 + wave;
    const long u0 = wg * U / nwv, u1 = (wg + 1) * U / nwv;
    const long t0 = u0 * 8;
    const int nvalid = (int)(u1 - u0) * 8;
    const char* wsrc = (const char*)p.wstream;
    auto dma1k = [&](const char* src_uniform, unsigned voff, unsigned lds_dst) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)src_uniform);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)src_uniform >> 32));
        const char* sp = (const char*)(((uintptr_t)hi << 32) | lo);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(sp) : "memory");
    };
    auto dma_piece = [&](int piece) {
        const unsigned slot = (unsigned)piece % NSLOT;
        const char* src = wsrc + (size_t)piece * PIECE_B;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const unsigned kb = (unsigned)(wave * DPW + i) * 1024u;
            dma1k(src, kb + lane16, lds0 + slot * PIECE_B + kb);
        }
    };
    int issued = 0;
    auto boundary = [&](int a, int b, int extra) {
        __builtin_amdgcn_sched_barrier(0);
        vb_wait_le<12>((issued - b) * DPW + extra);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (see vitblock_kernel's boundary)
        __builtin_amdgcn_s_barrier();
        int lim = a + NSLOT; lim = lim < NP ? lim : NP;
        while (issued < lim) { dma_piece(issued); ++issued; }
    };
    auto frag = [&](int piece, int f) -> V8 { return *(const V8*)(smem + ((unsigned)piece % NSLOT) * PIECE_B + f * 1024 + lane16); };
    {
        const char* vsrc = (const char*)p.vec;
#pragma unroll
        for (int i = 0; i < VEC_DPW; ++i) {
            const unsigned kb = (unsigned)(wave * VEC_DPW + i) * 1024u;
            dma1k(vsrc, kb + lane16, lds0 + NSLOT * PIECE_B + kb);
        }
        for (; issued < NSLOT && issued < NP; ++issued) dma_piece(issued);
    }
    const T* x_w = (const T*)p.x + t0 * p.ldx;
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)x_w, 0, (int)(nvalid * p.ldx * 2), 0x00020000);
    V8 xf[NH][KS];
#pragma unroll
    for (int th = 0; th < NH; ++th)
#pragma unroll
        for (int t = 0; t < KS; ++t)
            xf[th][t] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(r_x, (unsigned)(((32 * th + j) * p.ldx + 16 * t + 8 * h) * 2), 0, 0));
    boundary(0, 2, NH * KS);
    // ---- LayerNorm of the rows (two passes over the registers; lanes (j, 0) and (j, 1) hold alternating 8-channel runs of token j)
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += to_f32<T>(xf[th][t][e]);
        s += __shfl_xor(s, 32);
        const float mean = s * (1.f / C);
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dl = to_f32<T>(xf[th][t][e]) - mean; v = fmaf(dl, dl, v); }
        v += __shfl_xor(v, 32);
        const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps_next), nmr = -mean * rstd;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            u32x4 w;
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] = pack2<T>(fmaf(to_f32<T>(xf[th][t][2 * d]), rstd, nmr), fmaf(to_f32<T>(xf[th][t][2 * d + 1]), rstd, nmr));
            xf[th][t] = __builtin_bit_cast(V8, w);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    auto bias16 = [&](const float* src) -> f32x16 {
        f32x16 r;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 v = *(const f32x4*)(src + 8 * b + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[4 * b + e] = v[e];
        }
        return r;
    };
    const __amdgpu_buffer_rsrc_t r_q = __builtin_amdgcn_make_buffer_rsrc(p.q, 0, (int)p.qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_k = __builtin_amdgcn_make_buffer_rsrc(p.k, 0, (int)p.qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc(p.vt, 0, (int)p.qkv_bytes, 0x00020000);
    const int hd = 1 << p.hd_log2;
    unsigned row_qk[NH];
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        const unsigned tok = (unsigned)t0 + 32 * th + j;
        const unsigned img = tok / (unsigned)p.Tp, wi = tok - img * (unsigned)p.Tp;
        row_qk[th] = 32 * th + j < nvalid ? (unsigned)(((long)img * p.heads * p.Tp + wi) << p.hd_log2) : 0x7fffffffu;
    }
    constexpr int SPS = 2 * NH * 2, LAG = (NSLOT - 2) / 2;
    unsigned row_v8[NH][2];
#pragma unroll
    for (int th = 0; th < NH; ++th)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int tl = 32 * th + 16 * jb + 8 * h;
            const unsigned tk = (unsigned)t0 + tl;
            const unsigned im = tk / (unsigned)p.Tp, wv = tk - im * (unsigned)p.Tp;
            row_v8[th][jb] = tl < nvalid ? (unsigned)((long)im * p.heads * hd * p.Tp + wv) : 0x7fffffffu;
        }
#pragma unroll 1
    for (int s = 0; s < NP / 2; ++s) {
        if (s > 0) boundary(2 * s, 2 * s + 2, s < LAG ? SPS * s : SPS * LAG);
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pi = 2 * s + pp, piece = pi;
            const int sg = pi / NTI, nl0 = (pi - sg * NTI) * 32;
            f32x16 acc[NH];
            if (sg < 2) {
                const f32x16 bias = bias16(bqs + sg * C + nl0);
                V8 fr[RD];
#pragma unroll
                for (int i = 0; i < RD; ++i) fr[i] = frag(piece, i);
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    const V8 a = fr[t % RD];
#pragma unroll
                    for (int th = 0; th < NH; ++th) acc[th] = Mma32<T>::k16(a, xf[th][t], t == 0 ? bias : acc[th]);
                    if (t + RD < KS) fr[t % RD] = frag(piece, t + RD);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (sg == 0) {
#pragma unroll
                    for (int th = 0; th < NH; ++th)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[th][e] *= p.qscale;
                }
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const int f = nl0 + 16 * jb + 8 * h, hh = f >> p.hd_log2, dd = f & (hd - 1);
                    const unsigned col = (unsigned)(((long)hh * p.Tp << p.hd_log2) + dd);
#pragma unroll
                    for (int th = 0; th < NH; ++th) {
                        const u32x4 ow = vb_rows8(pack2<T>(acc[th][8 * jb], acc[th][8 * jb + 1]), pack2<T>(acc[th][8 * jb + 2], acc[th][8 * jb + 3]),
                                                  pack2<T>(acc[th][8 * jb + 4], acc[th][8 * jb + 5]), pack2<T>(acc[th][8 * jb + 6], acc[th][8 * jb + 7]));
                        const unsigned off = row_qk[th] == 0x7fffffffu ? 0x80000000u : (row_qk[th] + col) * 2u;
                        if (sg == 0) __builtin_amdgcn_raw_buffer_store_b128(ow, r_q, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(ow, r_k, off, 0, 0);
                    }
                }
            } else {
                const float bv = bqs[2 * C + nl0 + j];
                f32x16 binit;
#pragma unroll
                for (int e = 0; e < 16; ++e) binit[e] = bv;
                V8 fr[RD];
#pragma unroll
                for (int i = 0; i < RD; ++i) fr[i] = frag(piece, i);
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    const V8 a = fr[t % RD];
#pragma unroll
                    for (int th = 0; th < NH; ++th) acc[th] = Mma32<T>::k16(xf[th][t], a, t == 0 ? binit : acc[th]);
                    if (t + RD < KS) fr[t % RD] = frag(piece, t + RD);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int f = nl0 + j, hh = f >> p.hd_log2, dd = f & (hd - 1);
                const unsigned rowb = (unsigned)(((long)hh * hd + dd) * p.Tp);
#pragma unroll
                for (int th = 0; th < NH; ++th)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const u32x4 ow = vb_rows8(pack2<T>(acc[th][8 * jb], acc[th][8 * jb + 1]), pack2<T>(acc[th][8 * jb + 2], acc[th][8 * jb + 3]),
                                                  pack2<T>(acc[th][8 * jb + 4], acc[th][8 * jb + 5]), pack2<T>(acc[th][8 * jb + 6], acc[th][8 * jb + 7]));
                        const unsigned off = row_v8[th][jb] == 0x7fffffffu ? 0x80000000u : (row_v8[th][jb] + rowb) * 2u;
                        __builtin_amdgcn_raw_buffer_store_b128(ow, r_v, off, 0, 0);
                    }
            }
        }
    }
}

template <typename T, int C, int NH>
int launch_vq(const VbParams& p, hipStream_t st) {
    constexpr int KS = C / 16, PIECE_B = KS * 1024, NSLOT = C == 192 ? 8 : 5;
    constexpr int VEC_B = (3 * C * 4 + 4095) / 4096 * 4096;
    constexpr size_t lds = (size_t)NSLOT * PIECE_B + VEC_B;
    static bool attr_done[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)vit_qkv_kernel<T, C, NH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    const long per_wg = 4L * 32 * NH;
    long grid = (p.M + per_wg - 1) / per_wg;
    while (((p.M / 8 + grid * 4 - 1) / (grid * 4)) * 8 > 32 * NH) ++grid;
    ProfScope ps(KID_VITBLOCK, 6.0 * p.M * C * C, (double)p.M * C * sizeof(T) * 4, st);
    hipLaunchKernelGGL((vit_qkv_kernel<T, C, NH>), dim3((unsigned)grid), dim3(256), lds, st, p);
    return lwdetr_check_launch();
}


// ------------------------------------------------------------------------------------------- patch embedding + norm1 + QKV
// The ViT stem as one launch (round 4): x0 = patches Wpe^T + b + pos (vit.py:353-358: Conv2d(3, C, 16, stride 16) on the NCHW image
// + absolute position embedding), stored as the residual stream, then block 0's norm1 + QKV from the registers (vit.py:199,
// :123-130) - the patch GEMM (58 us at BASELINE config 2: a 64 x 64 tiling re-gathers every pixel for three column tiles) and
// lwdetr_vit_qkv (30 us) before. A wave owns 32 NH window-major tokens exactly as vitblock_kernel does; the contraction runs over
// k = (channel, patch row, pixel) in 48 steps of 16 pixels = ONE 32-byte patch row per token, which a lane pair reads straight from
// the image as the B operand (two steps of four rows ahead of the MFMAs that consume them: no LDS, every pixel read once). Wpe
// streams through the same ring as every other weight of the ViT: piece i = fragments (k-step 2 i + kk, channel tile n), kk-major.
// The accumulators start at pos + b; x0 is rounded to the storage type once (as the GEMM epilogue did), and the QKV phase is the
// one of vitblock_kernel (weights in accumulator k-slot order).
struct VsParams {
    const void* img; unsigned img_bytes; int img_h, img_w;     // (B, 3, H, W) of T
    int Hp, Wp, Twp;                                           // window-major token layout (common.h: tok_decode)
    const void* pos; long ldpos;                               // (Tp, C) of T: position embedding in token order (zero pad rows)
    VbParams v;                                                // x (out), wstream, vec, q / k / vt, M, eps_next, qscale, heads, hd_log2, Tp
};

template <typename T, int C, int NH>
__global__ __launch_bounds__(256, 1) void vit_stem_kernel(const VsParams ps) {
    typedef typename Vec<T>::v8 V8;
    static_assert(sizeof(T) == 2, "16-bit types only");
    const VbParams& p = ps.v;
    constexpr int KS = C / 16, NTI = C / 32;
    constexpr int PIECE_B = KS * 1024, DPW = KS / 4;
    constexpr int NSLOT = C == 192 ? 8 : 5;
    constexpr int VEC_B = (4 * C * 4 + 4095) / 4096 * 4096, VEC_DPW = VEC_B / 4096;
    constexpr int NP_PE = 24, NP_QKV = 3 * NTI, NP = NP_PE + NP_QKV, Q0 = NP_PE;
    constexpr int NSTEP = NP_PE / 2;            // ring steps of the patch phase: 2 pieces = 4 k-steps (patch rows) each
    constexpr int NPX = NH * 4;                 // pixel loads per step
    constexpr int PF = 2;                       // pixel loads run PF steps ahead (three register buffers)
    constexpr int RD = NH == 2 ? 4 : 8;
    static_assert(DPW % 3 == 0 && KS == 2 * NTI, "wait counts are kept in multiples of 3");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float* vec = (const float*)(smem + NSLOT * PIECE_B);
    const float* bpe = vec; const float* bqs = vec + C;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const long U = p.M >> 3, nwv = (long)gridDim.x * 4, wg = (long)blockIdx.x * 4 + wave;
    const long u0 = wg * U / nwv, u1 = (wg + 1) * U / nwv;
    const long t0 = u0 * 8;
    const int nvalid = (int)(u1 - u0) * 8;
    const char* wsrc = (const char*)p.wstream;
    auto dma1k = [&](const char* src_uniform, unsigned voff, unsigned lds_dst) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)src_uniform);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)src_uniform >> 32));
        const char* sp = (const char*)(((uintptr_t)hi << 32) | lo);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(sp) : "memory");
    };
    auto dma_piece = [&](int piece) {
        const unsigned slot = (unsigned)piece % NSLOT;
        const char* src = wsrc + (size_t)piece * PIECE_B;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const unsigned kb = (unsigned)(wave * DPW + i) * 1024u;
            dma1k(src, kb + lane16, lds0 + slot * PIECE_B + kb);
        }
    };
    int issued = 0;
    auto boundary = [&](int a, int b, int extra) {
        __builtin_amdgcn_sched_barrier(0);
        vb_wait_le<12>((issued - b) * DPW + extra);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (see vitblock_kernel's boundary)
        __builtin_amdgcn_s_barrier();
        int lim = a + NSLOT; lim = lim < NP ? lim : NP;
        while (issued < lim) { dma_piece(issued); ++issued; }
    };
    auto frag = [&](int piece, int f) -> V8 { return *(const V8*)(smem + ((unsigned)piece % NSLOT) * PIECE_B + f * 1024 + lane16); };
    {
        const char* vsrc = (const char*)p.vec;
#pragma unroll
        for (int i = 0; i < VEC_DPW; ++i) {
            const unsigned kb = (unsigned)(wave * VEC_DPW + i) * 1024u;
            dma1k(vsrc, kb + lane16, lds0 + NSLOT * PIECE_B + kb);
        }
        for (; issued < NSLOT; ++issued) dma_piece(issued);
    }
    // ---- this lane's tokens: byte offset of pixel (0, 8 h) of the patch in channel 0, row of the position table
    const __amdgpu_buffer_rsrc_t r_img = __builtin_amdgcn_make_buffer_rsrc((void*)ps.img, 0, (int)ps.img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_pos = __builtin_amdgcn_make_buffer_rsrc((void*)ps.pos, 0, (int)(p.Tp * ps.ldpos * 2), 0x00020000);
    T* x_w = (T*)p.x + t0 * p.ldx;
    const __amdgpu_buffer_rsrc_t r_x = __builtin_amdgcn_make_buffer_rsrc((void*)x_w, 0, (int)(nvalid * p.ldx * 2), 0x00020000);
    unsigned pix0[NH], prow[NH];
    {
        TokLayout L; L.winmajor = 1; L.Hp = ps.Hp; L.Wp = ps.Wp; L.Twp = ps.Twp;
#pragma unroll
        for (int th = 0; th < NH; ++th) {
            const long tok = t0 + 32 * th + j;
            const TokPos tp = tok_decode(tok, L);
            const bool ok = 32 * th + j < nvalid && tp.valid;
            pix0[th] = ok ? (unsigned)((((long)tp.b * 3 * ps.img_h + tp.y * 16) * ps.img_w + tp.x * 16 + 8 * h) * 2) : 0x80000000u;
            prow[th] = (unsigned)((tok % p.Tp) * ps.ldpos * 2);
        }
    }
    const unsigned plane_b = (unsigned)ps.img_h * (unsigned)ps.img_w * 2u, row_b = (unsigned)ps.img_w * 2u;
    V8 px[PF + 1][NH][4];                       // [buffer = step % 3][token half][patch row of the step]
    auto load_px = [&](auto step_tag) {
        constexpr int st = decltype(step_tag)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            constexpr int dummy = 0; (void)dummy;
            const int t = 4 * st + kk, ch = t >> 4, py = t & 15;                 // compile-time after unrolling
            const unsigned so = __builtin_amdgcn_readfirstlane(ch * plane_b + py * row_b);
#pragma unroll
            for (int th = 0; th < NH; ++th)
                px[st % (PF + 1)][th][kk] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(r_img, pix0[th], so, 0));
        }
    };
    // position rows (16-byte loads, channels 32 n + 16 jb + 8 h .. + 7 of the token), then the pixels of steps 0 and 1
    f32x16 acc2[NTI][NH];
    {
        u32x4 pv[NH][NTI][2];
#pragma unroll
        for (int th = 0; th < NH; ++th)
#pragma unroll
            for (int n = 0; n < NTI; ++n)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
                    pv[th][n][jb] = __builtin_amdgcn_raw_buffer_load_b128(r_pos, prow[th] + (unsigned)((32 * n + 16 * jb + 8 * h) * 2), 0, 0);
        load_px(std::integral_constant<int, 0>{});
        load_px(std::integral_constant<int, 1>{});
        boundary(0, 2, NH * NTI * 2 + 2 * NPX);
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            f32x16 na[NH];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int th = 0; th < NH; ++th) {
                    const u32x4 own = vb_rows8(pv[th][n][jb][0], pv[th][n][jb][1], pv[th][n][jb][2], pv[th][n][jb][3]);
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int b = 2 * jb + bb, c0 = 32 * n + 8 * b + 4 * h;
                        const f32x4 bbv = *(const f32x4*)(bpe + c0);
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const unsigned ow_ = own[2 * bb + d];
                            const typename Pk<T>::v2 v2 = __builtin_bit_cast(typename Pk<T>::v2, ow_);
                            na[th][4 * b + 2 * d] = to_f32<T>(v2[0]) + bbv[2 * d];
                            na[th][4 * b + 2 * d + 1] = to_f32<T>(v2[1]) + bbv[2 * d + 1];
                        }
                    }
                }
#pragma unroll
            for (int th = 0; th < NH; ++th) {
                asm volatile("" : "+a"(na[th]));
                acc2[n][th] = na[th];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- patch phase: step s = pieces 2 s, 2 s + 1 = patch rows 4 s .. 4 s + 3, fragments (row, channel tile) in stream order.
    // Vector-memory operations younger than a step's pieces besides later pieces (exact; the compiler's own waits for the pixel
    // registers see only its loads and can only wait longer): the position loads and every pixel group issued since.
    vb_static_for<NSTEP>([&](auto s_tag) {
        constexpr int s = decltype(s_tag)::value;
        if constexpr (s > 0) {
            constexpr int npos = s < 4 ? NH * NTI * 2 : 0;
            constexpr int hi = s + 1 < NSTEP - 1 ? s + 1 : NSTEP - 1;           // youngest pixel group issued before this boundary
            constexpr int lo = s < 4 ? 0 : s - 1;                               // oldest one younger than piece 2 s + 1
            boundary(2 * s, 2 * s + 2, npos + (hi - lo + 1) * NPX);
        }
        if constexpr (s + PF < NSTEP) load_px(std::integral_constant<int, s + PF>{});
        V8 fr[RD];
#pragma unroll
        for (int i = 0; i < RD; ++i) fr[i] = frag(2 * s + i / KS, i % KS);
#pragma unroll
        for (int fi = 0; fi < 2 * KS; ++fi) {
            const int kk = fi / NTI, n = fi % NTI;
            const V8 a = fr[fi % RD];
#pragma unroll
            for (int th = 0; th < NH; ++th) acc2[n][th] = Mma32<T>::k16(a, px[s % (PF + 1)][th][kk], acc2[n][th]);
            if (fi + RD < 2 * KS) fr[fi % RD] = frag(2 * s + (fi + RD) / KS, (fi + RD) % KS);
            __builtin_amdgcn_sched_barrier(0);
        }
    });
    // ---- x0 rounded to the storage type and stored; its statistics; LN(x0) as the B operand of the QKV phase
    V8 xf[NH][KS];
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        float sm = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                unsigned pw[4];
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int b = 2 * jb + bb;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const unsigned w = pack2<T>(acc2[n][th][4 * b + 2 * d], acc2[n][th][4 * b + 2 * d + 1]);
                        pw[2 * bb + d] = w;
                        const typename Pk<T>::v2 v2 = __builtin_bit_cast(typename Pk<T>::v2, w);
                        const float v0 = to_f32<T>(v2[0]), v1 = to_f32<T>(v2[1]);
                        acc2[n][th][4 * b + 2 * d] = v0; acc2[n][th][4 * b + 2 * d + 1] = v1;
                        sm += v0 + v1;
                    }
                }
                const u32x4 ow = vb_rows8(pw[0], pw[1], pw[2], pw[3]);
                __builtin_amdgcn_raw_buffer_store_b128(ow, r_x, (unsigned)(((32 * th + j) * p.ldx + 32 * n + 16 * jb + 8 * h) * 2), 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        sm += __shfl_xor(sm, 32);
        const float mean = sm * (1.f / C);
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float dl = acc2[n][th][e] - mean; v += dl * dl; }
        v += __shfl_xor(v, 32);
        const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps_next), nmr = -mean * rstd;
#pragma unroll
        for (int n = 0; n < NTI; ++n)
#pragma unroll
            for (int be = 0; be < 2; ++be) {
                u32x4 w;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int r = (2 * be + (d >> 1)) * 4 + (d & 1) * 2;
                    w[d] = pack2<T>(fmaf(acc2[n][th][r], rstd, nmr), fmaf(acc2[n][th][r + 1], rstd, nmr));
                }
                xf[th][2 * n + be] = __builtin_bit_cast(V8, w);
            }
    }
    auto bias16 = [&](const float* src) -> f32x16 {
        f32x16 r;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 v = *(const f32x4*)(src + 8 * b + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[4 * b + e] = v[e];
        }
        return r;
    };
    // ---- norm1 + QKV of block 0: as the chained phase of vitblock_kernel
    const __amdgpu_buffer_rsrc_t r_q = __builtin_amdgcn_make_buffer_rsrc(p.q, 0, (int)p.qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_k = __builtin_amdgcn_make_buffer_rsrc(p.k, 0, (int)p.qkv_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc(p.vt, 0, (int)p.qkv_bytes, 0x00020000);
    const int hd = 1 << p.hd_log2;
    unsigned row_qk[NH];
#pragma unroll
    for (int th = 0; th < NH; ++th) {
        const unsigned tok = (unsigned)t0 + 32 * th + j;
        const unsigned img = tok / (unsigned)p.Tp, wi = tok - img * (unsigned)p.Tp;
        row_qk[th] = 32 * th + j < nvalid ? (unsigned)(((long)img * p.heads * p.Tp + wi) << p.hd_log2) : 0x7fffffffu;
    }
    constexpr int XST = NH * NTI * 2, SPS = 2 * NH * 2, LAG = (NSLOT - 2) / 2;
    unsigned row_v8[NH][2];
#pragma unroll
    for (int th = 0; th < NH; ++th)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int tl = 32 * th + 16 * jb + 8 * h;
            const unsigned tk = (unsigned)t0 + tl;
            const unsigned im = tk / (unsigned)p.Tp, wv = tk - im * (unsigned)p.Tp;
            row_v8[th][jb] = tl < nvalid ? (unsigned)((long)im * p.heads * hd * p.Tp + wv) : 0x7fffffffu;
        }
#pragma unroll 1
    for (int s = 0; s < NP_QKV / 2; ++s) {
        boundary(Q0 + 2 * s, Q0 + 2 * s + 2, s < LAG ? XST + SPS * s : SPS * LAG);
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pi = 2 * s + pp, piece = Q0 + pi;
            const int sg = pi / NTI, nl0 = (pi - sg * NTI) * 32;
            f32x16 acc[NH];
            if (sg < 2) {
                const f32x16 bias = bias16(bqs + sg * C + nl0);
                V8 fr[RD];
#pragma unroll
                for (int i = 0; i < RD; ++i) fr[i] = frag(piece, i);
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    const V8 a = fr[t % RD];
#pragma unroll
                    for (int th = 0; th < NH; ++th) acc[th] = Mma32<T>::k16(a, xf[th][t], t == 0 ? bias : acc[th]);
                    if (t + RD < KS) fr[t % RD] = frag(piece, t + RD);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (sg == 0) {
#pragma unroll
                    for (int th = 0; th < NH; ++th)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[th][e] *= p.qscale;
                }
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const int f = nl0 + 16 * jb + 8 * h, hh = f >> p.hd_log2, dd = f & (hd - 1);
                    const unsigned col = (unsigned)(((long)hh * p.Tp << p.hd_log2) + dd);
#pragma unroll
                    for (int th = 0; th < NH; ++th) {
                        const u32x4 ow = vb_rows8(pack2<T>(acc[th][8 * jb], acc[th][8 * jb + 1]), pack2<T>(acc[th][8 * jb + 2], acc[th][8 * jb + 3]),
                                                  pack2<T>(acc[th][8 * jb + 4], acc[th][8 * jb + 5]), pack2<T>(acc[th][8 * jb + 6], acc[th][8 * jb + 7]));
                        const unsigned off = row_qk[th] == 0x7fffffffu ? 0x80000000u : (row_qk[th] + col) * 2u;
                        if (sg == 0) __builtin_amdgcn_raw_buffer_store_b128(ow, r_q, off, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(ow, r_k, off, 0, 0);
                    }
                }
            } else {
                const float bv = bqs[2 * C + nl0 + j];
                f32x16 binit;
#pragma unroll
                for (int e = 0; e < 16; ++e) binit[e] = bv;
                V8 fr[RD];
#pragma unroll
                for (int i = 0; i < RD; ++i) fr[i] = frag(piece, i);
#pragma unroll
                for (int t = 0; t < KS; ++t) {
                    const V8 a = fr[t % RD];
#pragma unroll
                    for (int th = 0; th < NH; ++th) acc[th] = Mma32<T>::k16(xf[th][t], a, t == 0 ? binit : acc[th]);
                    if (t + RD < KS) fr[t % RD] = frag(piece, t + RD);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int f = nl0 + j, hh = f >> p.hd_log2, dd = f & (hd - 1);
                const unsigned rowb = (unsigned)(((long)hh * hd + dd) * p.Tp);
#pragma unroll
                for (int th = 0; th < NH; ++th)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const u32x4 ow = vb_rows8(pack2<T>(acc[th][8 * jb], acc[th][8 * jb + 1]), pack2<T>(acc[th][8 * jb + 2], acc[th][8 * jb + 3]),
                                                  pack2<T>(acc[th][8 * jb + 4], acc[th][8 * jb + 5]), pack2<T>(acc[th][8 * jb + 6], acc[th][8 * jb + 7]));
                        const unsigned off = row_v8[th][jb] == 0x7fffffffu ? 0x80000000u : (row_v8[th][jb] + rowb) * 2u;
                        __builtin_amdgcn_raw_buffer_store_b128(ow, r_v, off, 0, 0);
                    }
            }
        }
    }
}

template <typename T, int C, int NH>
int launch_vs(const VsParams& p, hipStream_t st) {
    constexpr int KS = C / 16, PIECE_B = KS * 1024, NSLOT = C == 192 ? 8 : 5;
    constexpr int VEC_B = (4 * C * 4 + 4095) / 4096 * 4096;
    constexpr size_t lds = (size_t)NSLOT * PIECE_B + VEC_B;
    static bool attr_done[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)vit_stem_kernel<T, C, NH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    const long M = p.v.M, per_wg = 4L * 32 * NH;
    long grid = (M + per_wg - 1) / per_wg;
    while (((M / 8 + grid * 4 - 1) / (grid * 4)) * 8 > 32 * NH) ++grid;
    ProfScope ps(KID_VITBLOCK, (2.0 * 768 + 6.0 * C) * M * C, (double)M * 768 * sizeof(T) + (double)M * C * sizeof(T) * 5, st);
    hipLaunchKernelGGL((vit_stem_kernel<T, C, NH>), dim3((unsigned)grid), dim3(256), lds, st, p);
    return lwdetr_check_launch();
}

struct VbLaunchState { bool attr_done; int ncu; };

template <typename T, int C, int NH, bool QKV, int WPC = 1, bool G16 = false>
int launch_vb(const VbParams& p, hipStream_t st) {
    constexpr int KS = C / 16, PIECE_B = KS * 1024, NSLOT = WPC == 2 ? 5 : (C == 192 ? 8 : 5);
    constexpr int VEC_B = (13 * C * 4 + 4095) / 4096 * 4096;
    constexpr size_t lds = (size_t)NSLOT * PIECE_B + VEC_B;
    static VbLaunchState state[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    VbLaunchState& s = state[dev];
    if (!s.attr_done) {
        if (hipFuncSetAttribute((const void*)vitblock_kernel<T, C, NH, QKV, WPC, G16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LWDETR_ERR_LAUNCH;
        s.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        s.attr_done = true;
    }
    // grid: whole rounds of one workgroup per CU, tokens dealt evenly (every wave at most 32 NH tokens)
    const long per_wg = 4L * 32 * NH;
    const long need = (p.M + per_wg - 1) / per_wg;
    // Full tiles on as few workgroups as the rows need (a tile costs the same matrix time however many of its 32 NH token slots
    // are filled, so spreading 200 workgroups' rows over 256 buys nothing and takes CUs from whatever runs beside this launch);
    // LWDETR_VB_GRID=rounds restores whole rounds of one workgroup per CU (round-3 measurements), a number forces the grid.
    const long gknob = lwdetr_knob(KNOB_VB_GRID, 0);       // -1 = "rounds"
    long grid = need;
    if (gknob == -1) grid = (need + s.ncu - 1) / s.ncu * s.ncu;
    else if (gknob >= need) grid = gknob;
    // 8-token units are dealt by floor(): a wave can get one unit more than the average
    while (((p.M / 8 + grid * 4 - 1) / (grid * 4)) * 8 > 32 * NH) ++grid;
    ProfScope ps(KID_VITBLOCK, (16.0 + 2.0 + (QKV ? 6.0 : 0.0)) * p.M * C * C,
                 (double)p.M * C * sizeof(T) * 3 + (QKV ? 3.0 : 0.0) * p.M * C * sizeof(T) + (p.out2 ? 1.0 : 0.0) * p.M * C * sizeof(T), st);
    hipLaunchKernelGGL((vitblock_kernel<T, C, NH, QKV, WPC, G16>), dim3((unsigned)grid), dim3(256), lds, st, p);
    return lwdetr_check_launch();
}

template <typename T>
int dispatch_vb(const VbParams& p, int C, bool qkv, hipStream_t st) {
    // C = 192, 32 tokens per wave (128 per workgroup, <= 256 registers, 72 KB of LDS: two workgroups per CU) while all of the launch's
    // workgroups are resident at once (M <= 65 536 rows on 256 CUs). Round 5, profiles/r5b_vitblock_half_tiles.txt: the launch's time is
    // one workgroup's latency either way - 84 -> 67 us at M = 25 600 (one launch chain of config 2: 200 workgroups, one per CU), equal at
    // M = 51 200 (400 workgroups, two per CU: 93.7 vs 92 us) - and its waves leave half of each SIMD's registers to the other chain's
    // kernels: config 2 +1.3 % (two A/B pairs on one box). LWDETR_VB_HALF=0|1 forces either form (read per launch: tests switch it).
    // f16: the GELU on packed f16 pairs (the helpers' comment above; profiles/r5h_block_kernel_packed_f16_gelu.txt: launch -1.5 ... -2.5 %,
    // config 2 / large +0.8 / +0.9 %, the model's mean 16-bit error +0.9 %). LWDETR_VB_GELU16=0 restores the f32-arithmetic form (read per
    // launch: tests switch it); bf16 has no packed arithmetic and keeps it.
    const bool g16 = std::is_same<T, f16>::value && lwdetr_knob(KNOB_VB_GELU16, 1) == 1;
    if (C == 192) {
        int dev = 0; (void)hipGetDevice(&dev);
        static int ncu[16] = {};
        if (dev >= 0 && dev < 16 && ncu[dev] == 0) { hipDeviceProp_t prop; ncu[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
        const long cus = dev >= 0 && dev < 16 ? ncu[dev] : 256;
        const bool half = lwdetr_knob_is_set(KNOB_VB_HALF) ? lwdetr_knob(KNOB_VB_HALF, 0) == 1 : (p.M + 127) / 128 <= 2 * cus;
        if (half) {
            if constexpr (sizeof(T) == 2 && std::is_same<T, f16>::value)
                if (g16) return qkv ? launch_vb<T, 192, 1, true, 2, true>(p, st) : launch_vb<T, 192, 1, false, 2, true>(p, st);
            return qkv ? launch_vb<T, 192, 1, true, 2>(p, st) : launch_vb<T, 192, 1, false, 2>(p, st);
        }
    }
    if constexpr (std::is_same<T, f16>::value) {
        if (g16 && C == 192) return qkv ? launch_vb<T, 192, 2, true, 1, true>(p, st) : launch_vb<T, 192, 2, false, 1, true>(p, st);
        if (g16 && C == 384) return qkv ? launch_vb<T, 384, 1, true, 1, true>(p, st) : launch_vb<T, 384, 1, false, 1, true>(p, st);
    }
    if (C == 192) return qkv ? launch_vb<T, 192, 2, true>(p, st) : launch_vb<T, 192, 2, false>(p, st);
    if (C == 384) return qkv ? launch_vb<T, 384, 1, true>(p, st) : launch_vb<T, 384, 1, false>(p, st);
    return LWDETR_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" long lwdetr_vit_block_stream_bytes(int C, int has_qkv) {
    if (C != 192 && C != 384) return -LWDETR_ERR_UNSUPPORTED;
    const long np = C / 32 + 2 * (C / 8) + (has_qkv ? 3 * (C / 32) : 0);
    return np * (C / 16) * 1024L;
}
extern "C" long lwdetr_vit_block_vec_floats(int C) { return ((13L * C * 4 + 4095) / 4096 * 4096) / 4; }

extern "C" int lwdetr_vit_block(void* x, long ldx, const void* att, long ldatt, const void* wstream, const float* vec, void* out2,
                                long ld2, float* stats_out, long M, int C, float eps, float eps_next, int has_qkv, void* q_out,
                                void* k_out, void* vt_out, float qscale, int heads, int hd, int Tp, int dtype, void* hip_stream) {
    if (!x || !att || !wstream || !vec || M < 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    if (ldx % 8 != 0 || ldatt % 8 != 0 || (out2 && ld2 % 8 != 0)) return LWDETR_ERR_BAD_ARG;
    if (M % 8 != 0) return LWDETR_ERR_UNSUPPORTED;        // tokens are dealt to waves in runs of 8 (16-byte V^T stores)
    if (((uintptr_t)wstream | (uintptr_t)vec | (uintptr_t)att) % 16 != 0 || (uintptr_t)x % 16 != 0 || (uintptr_t)out2 % 16 != 0) return LWDETR_ERR_BAD_ARG;
    VbParams p = {};
    p.x = x; p.ldx = ldx; p.att = att; p.ldatt = ldatt; p.wstream = wstream; p.vec = vec; p.out2 = out2; p.ld2 = ld2;
    p.stats_out = stats_out; p.M = M; p.eps = eps; p.eps_next = eps_next; p.qscale = qscale;
    if (has_qkv) {
        if (!q_out || !k_out || !vt_out || heads <= 0 || hd < 4 || (hd & (hd - 1)) != 0 || heads * hd != C || Tp <= 0 || Tp % 4 != 0 ||
            ((uintptr_t)q_out | (uintptr_t)k_out | (uintptr_t)vt_out) % 16 != 0)
            return LWDETR_ERR_BAD_ARG;
        if (hd < 8 || Tp % 8 != 0) return LWDETR_ERR_UNSUPPORTED;
        if (M % Tp != 0 || (double)M * C * 2.0 >= 2147483000.0) return LWDETR_ERR_UNSUPPORTED;
        int l2 = 0; while ((1 << l2) < hd) ++l2;
        p.q = q_out; p.k = k_out; p.vt = vt_out; p.heads = heads; p.hd_log2 = l2; p.Tp = Tp;
        p.qkv_bytes = (unsigned)((unsigned long)M * C * 2ul);
    }
    // row offsets inside a wave's tile are 32-bit
    if ((double)(ldx > ldatt ? ldx : ldatt) * 64 * 2 >= 2147483000.0 || (out2 && (double)ld2 * 64 * 2 >= 2147483000.0)) return LWDETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F16: return dispatch_vb<f16>(p, C, has_qkv != 0, st);
        case DT_BF16: return dispatch_vb<bf16>(p, C, has_qkv != 0, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

extern "C" long lwdetr_vit_qkv_stream_bytes(int C) { return (C == 192 || C == 384) ? 3L * (C / 32) * (C / 16) * 1024L : -LWDETR_ERR_UNSUPPORTED; }
extern "C" long lwdetr_vit_qkv_vec_floats(int C) { return ((3L * C * 4 + 4095) / 4096 * 4096) / 4; }

extern "C" int lwdetr_vit_qkv(const void* x, long ldx, const void* wstream, const float* vec, long M, int C, float eps, void* q_out,
                              void* k_out, void* vt_out, float qscale, int heads, int hd, int Tp, int dtype, void* hip_stream) {
    if (!x || !wstream || !vec || !q_out || !k_out || !vt_out || M < 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    if (ldx % 8 != 0 || ((uintptr_t)x | (uintptr_t)wstream | (uintptr_t)vec | (uintptr_t)q_out | (uintptr_t)k_out | (uintptr_t)vt_out) % 16 != 0) return LWDETR_ERR_BAD_ARG;
    if (heads <= 0 || hd < 8 || (hd & (hd - 1)) != 0 || heads * hd != C || Tp <= 0 || Tp % 8 != 0) return LWDETR_ERR_UNSUPPORTED;
    if (M % 8 != 0 || M % Tp != 0 || (double)M * C * 2.0 >= 2147483000.0 || (double)ldx * 64 * 2 >= 2147483000.0) return LWDETR_ERR_UNSUPPORTED;
    VbParams p = {};
    p.x = (void*)x; p.ldx = ldx; p.wstream = wstream; p.vec = vec; p.M = M; p.eps_next = eps; p.qscale = qscale;
    int l2 = 0; while ((1 << l2) < hd) ++l2;
    p.q = q_out; p.k = k_out; p.vt = vt_out; p.heads = heads; p.hd_log2 = l2; p.Tp = Tp;
    p.qkv_bytes = (unsigned)((unsigned long)M * C * 2ul);
    hipStream_t st = (hipStream_t)hip_stream;
    if (C == 192) return dtype == DT_F16 ? launch_vq<f16, 192, 2>(p, st) : dtype == DT_BF16 ? launch_vq<bf16, 192, 2>(p, st) : LWDETR_ERR_UNSUPPORTED;
    if (C == 384) return dtype == DT_F16 ? launch_vq<f16, 384, 1>(p, st) : dtype == DT_BF16 ? launch_vq<bf16, 384, 1>(p, st) : LWDETR_ERR_UNSUPPORTED;
    return LWDETR_ERR_UNSUPPORTED;
}

extern "C" long lwdetr_vit_stem_stream_bytes(int C) { return (C == 192 || C == 384) ? (24L + 3L * (C / 32)) * (C / 16) * 1024L : -LWDETR_ERR_UNSUPPORTED; }
extern "C" long lwdetr_vit_stem_vec_floats(int C) { return ((4L * C * 4 + 4095) / 4096 * 4096) / 4; }

extern "C" int lwdetr_vit_stem(const void* img, int B, int img_h, int img_w, int Hp, int Wp, int Twp, const void* pos, long ldpos, void* x,
                               long ldx, const void* wstream, const float* vec, long M, int C, float eps, void* q_out, void* k_out,
                               void* vt_out, float qscale, int heads, int hd, int dtype, void* hip_stream) {
    if (!img || !pos || !x || !wstream || !vec || !q_out || !k_out || !vt_out || M < 0 || B <= 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    const int Tp = 16 * Twp;
    if (img_h != 16 * Hp || img_w != 16 * Wp || Hp % 4 != 0 || Wp % 4 != 0 || Twp < (Hp / 4) * (Wp / 4) || M != (long)B * Tp) return LWDETR_ERR_BAD_ARG;
    if (ldx % 8 != 0 || ldpos % 8 != 0 ||
        ((uintptr_t)img | (uintptr_t)pos | (uintptr_t)x | (uintptr_t)wstream | (uintptr_t)vec | (uintptr_t)q_out | (uintptr_t)k_out | (uintptr_t)vt_out) % 16 != 0)
        return LWDETR_ERR_BAD_ARG;
    if (heads <= 0 || hd < 8 || (hd & (hd - 1)) != 0 || heads * hd != C || Tp % 8 != 0) return LWDETR_ERR_UNSUPPORTED;
    if ((double)M * C * 2.0 >= 2147483000.0 || (double)ldx * 64 * 2 >= 2147483000.0 || (double)B * 3 * img_h * img_w * 2.0 >= 2147483000.0 ||
        (double)Tp * ldpos * 2.0 >= 2147483000.0)
        return LWDETR_ERR_UNSUPPORTED;
    VsParams p = {};
    p.img = img; p.img_bytes = (unsigned)((unsigned long)B * 3ul * img_h * img_w * 2ul); p.img_h = img_h; p.img_w = img_w;
    p.Hp = Hp; p.Wp = Wp; p.Twp = Twp; p.pos = pos; p.ldpos = ldpos;
    p.v.x = x; p.v.ldx = ldx; p.v.wstream = wstream; p.v.vec = vec; p.v.M = M; p.v.eps_next = eps; p.v.qscale = qscale;
    int l2 = 0; while ((1 << l2) < hd) ++l2;
    p.v.q = q_out; p.v.k = k_out; p.v.vt = vt_out; p.v.heads = heads; p.v.hd_log2 = l2; p.v.Tp = Tp;
    p.v.qkv_bytes = (unsigned)((unsigned long)M * C * 2ul);
    hipStream_t st = (hipStream_t)hip_stream;
    if (C == 192) return dtype == DT_F16 ? launch_vs<f16, 192, 2>(p, st) : dtype == DT_BF16 ? launch_vs<bf16, 192, 2>(p, st) : LWDETR_ERR_UNSUPPORTED;
    if (C == 384) return dtype == DT_F16 ? launch_vs<f16, 384, 1>(p, st) : dtype == DT_BF16 ? launch_vs<bf16, 384, 1>(p, st) : LWDETR_ERR_UNSUPPORTED;
    return LWDETR_ERR_UNSUPPORTED;
}
