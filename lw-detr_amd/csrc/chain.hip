// Row-local chains of Linear (+ activation / LayerNorm / row max) stages in ONE launch, for gfx950, 16-bit types (round 4).
//
// The LW-DETR forward has several runs of small GEMMs that never mix rows: the projector's last 1x1 convolution + LayerNorm, the
// two-stage head over all S encoder tokens (enc_output Linear -> LayerNorm -> class Linear -> row max; transformer.py:231-246), the
// value projections of every decoder layer (ms_deform_attn.py:110-114), the decoder's output projections + LayerNorms, the
// box / class heads. As separate launches each is a latency-bound 64 x 64 ring GEMM (0.21 of the HBM roofline, MFMA busy 0.06,
// profiles/r3_*) that writes its (rows x 256) activation to HBM for the next one to read back. Here a wave owns 32 rows for the whole
// chain, exactly as in vitblock.hip:
//   * D[channel][row] = W * x^T on 32x32x16 MFMAs: the weights are the A operand, the wave's rows the B operand, held in registers
//     as k-runs of 8; an accumulator tile handed on as the next B operand needs no data movement (its k-slot order is baked into
//     the packed weights, lwdetr_amd/kernels.py:chain_kslots); LayerNorm of a row = in-lane sums + one half-wave exchange;
//   * all weights of the chain are ONE stream of 4 KB pieces (32 output channels x 64 k-slots = 4 MFMA fragments of 1 KB in lane
//     order), DMA'd linearly through an LDS ring shared by the 4 waves of the workgroup (counted s_waitcnt vmcnt + one raw barrier
//     per output tile), fragment reads are base + lane * 16 (+ wrapped piece offset): conflict-free, no swizzle.
// Rounding points are those of the unfused launches (every stage output is rounded to the storage type before it is used again).
#include "common.h"
#include <cstdlib>
#include <cstring>

// Phase timing for tuning (tools/chain_timing.py builds a private copy with -DLWDETR_CH_TIMING; never in the product library):
// s_memrealtime stamps (10 ns ticks) of every wave of the first and the last workgroup of lwdetr_enc_chain at the stage boundaries,
// plus the time spent in the ring waits and barriers.
#ifdef LWDETR_CH_TIMING
__device__ unsigned long long g_ch_timing[2][4][16];
#define CH_TS(i) do { if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane == 0) g_ch_timing[blockIdx.x != 0][wave][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int lwdetr_debug_ch_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ch_timing), sizeof(g_ch_timing)) == hipSuccess ? 0 : 1;
}
#else
#define CH_TS(i) do {} while (0)
#endif

namespace {

template <typename T> struct Mma32c;
template <> struct Mma32c<f16> {
    static __device__ __forceinline__ f32x16 k16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mma32c<bf16> {
    static __device__ __forceinline__ f32x16 k16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
typedef unsigned int cu32x4 __attribute__((ext_vector_type(4)));
typedef float cf32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 cf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
template <typename T> struct CPk;
template <> struct CPk<f16> { typedef cf16x2 v2; };
template <> struct CPk<bf16> { typedef cbf16x2 v2; };
template <typename T> __device__ __forceinline__ unsigned cpack2(float a, float b) {
    const cf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, typename CPk<T>::v2));
}
template <typename T> __device__ __forceinline__ void cunpack2(unsigned w, float& a, float& b) {
    const typename CPk<T>::v2 v = __builtin_bit_cast(typename CPk<T>::v2, w);
    a = to_f32<T>(v[0]); b = to_f32<T>(v[1]);
}
// accumulator registers 8 jb .. 8 jb + 7 of lane (j, h) = rows 16 jb + 4 h + {0..3} and 16 jb + 8 + 4 h + {0..3} of column j, as
// four packed pairs; after the half-wave exchange every lane holds 8 CONSECUTIVE rows 16 jb + 8 h .. + 7 (one 16-byte store)
__device__ __forceinline__ cu32x4 crows8(unsigned a0, unsigned a1, unsigned b0, unsigned b1) {
    const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    return cu32x4{s0[0], s1[0], s0[1], s1[1]};
}

#define CH_VMW(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
// at most n vector-memory operations of this wave outstanding (n wave-uniform, 0 <= n <= 31: the ring never holds more pieces)
__device__ __forceinline__ void ch_wait_le(int n) {
    switch (n) {
#define CH_C(N) case N: CH_VMW(N); break;
        CH_C(1) CH_C(2) CH_C(3) CH_C(4) CH_C(5) CH_C(6) CH_C(7) CH_C(8) CH_C(9) CH_C(10) CH_C(11) CH_C(12) CH_C(13) CH_C(14) CH_C(15)
        CH_C(16) CH_C(17) CH_C(18) CH_C(19) CH_C(20) CH_C(21) CH_C(22) CH_C(23) CH_C(24) CH_C(25) CH_C(26) CH_C(27) CH_C(28) CH_C(29) CH_C(30)
        CH_C(31)
#undef CH_C
        default: CH_VMW(0); break;
    }
}

constexpr int CH_PIECE_B = 4096;          // 32 output channels x 64 k-slots: 4 fragments
constexpr int CH_RD = 8;                  // fragment read-ahead (registers); the host appends 2 zero pieces so that it may run past the end

// The weight stream of one workgroup: ring of NSLOT pieces in LDS, every wave DMAs fragment `wave` of every piece.
template <int NSLOT>
struct WRing {
    const char* src; unsigned lds0; int np; int issued; int wave; unsigned lane16;
    __device__ __forceinline__ void dma1k(const char* src_uniform, unsigned voff, unsigned lds_dst) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)src_uniform);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)src_uniform >> 32));
        const char* sp = (const char*)(((uintptr_t)hi << 32) | lo);
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(sp) : "memory");
    }
    __device__ __forceinline__ void dma_piece(int piece) {
        const unsigned slot = (unsigned)piece % NSLOT;
        dma1k(src + (size_t)piece * CH_PIECE_B, (unsigned)wave * 1024u + lane16, lds0 + slot * CH_PIECE_B + (unsigned)wave * 1024u);
    }
    __device__ __forceinline__ void fill(int first) { for (; issued < first && issued < NSLOT && issued < np; ++issued) dma_piece(issued); }
    // The tile that starts at piece a and spans n pieces: pieces below a + n + 2 have landed (the fragment read-ahead runs CH_RD = 8
    // fragments = 2 pieces ahead), every wave is done with the pieces below a, and their slots are refilled.
    // STEADY: the wait count while the ring is being topped up (NSLOT - pieces of the tile before - n - 2), one s_waitcnt without
    // the branch tree of the general case (which only the first and the last tiles of the stream take)
    template <int STEADY>
    __device__ __forceinline__ void begin_tile(int a, int n) {
        __builtin_amdgcn_sched_barrier(0);
        int b = a + n + 2; b = b < np ? b : np;
        const int v = issued - b;
        if (v == STEADY) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(STEADY) : "memory");
        else ch_wait_le(v);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of the slots that are about to be freed
        __builtin_amdgcn_s_barrier();
        int lim = a + NSLOT; lim = lim < np ? lim : np;
        while (issued < lim) { dma_piece(issued); ++issued; }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- exact form (enc_chain_kernel; measured with the form above: 0.4 us of every 0.8 us tile in begin_tile - the counted wait
    // took the tile epilogues' stores for prefetches still in flight, and four DMA issues of ~80 cycles each sat in front of the MFMAs;
    // profiles/r4c_enc_chain_phase_timing.txt). Every vector-memory operation of the wave gets a sequence number; seq_of[slot] is the
    // number of the DMA that filled the slot, so "piece q has landed" is vmcnt <= vmseq - seq_of[q]: stores never count as prefetches.
    // The refill of the freed slots is handed out one DMA per few MFMAs (issue_one) instead of a burst.
    int vmseq, need, lim; unsigned seq_lds;
#ifdef LWDETR_CH_TIMING
    unsigned long long tt_wait = 0, tt_bar = 0;
#endif
    __device__ __forceinline__ void seq_init(unsigned table_lds_byte_offset) {      // after a vmcnt(0): everything issued so far has landed
        vmseq = 0; need = 0; lim = issued; seq_lds = table_lds_byte_offset;
        for (int i = 0; i < NSLOT; ++i) *(__attribute__((address_space(3))) int*)(uintptr_t)(seq_lds + 4 * i) = 0;
    }
    __device__ __forceinline__ void issue_one() {
        if (issued < lim) {
            dma_piece(issued);
            ++vmseq;
            *(__attribute__((address_space(3))) int*)(uintptr_t)(seq_lds + 4 * ((unsigned)issued % NSLOT)) = vmseq;
            ++issued;
        }
    }
    __device__ __forceinline__ void flush() { while (issued < lim) issue_one(); }
    __device__ __forceinline__ void stores(int k) { vmseq += k; }      // k vector stores issued by this wave (never more than were issued)
    // n_next: pieces of the tile after this one (its wait count is fetched now)
    __device__ __forceinline__ void begin_tile_x(int a, int n, int n_next) {
        __builtin_amdgcn_sched_barrier(0);
        flush();
#ifdef LWDETR_CH_TIMING
        const unsigned long long ta = __builtin_amdgcn_s_memrealtime();
#endif
        int allowed = vmseq - need;                  // operations issued after the DMA of the last piece this tile reads ahead into
        allowed = allowed > 60 ? 60 : allowed & ~3;  // rounded down: a smaller branch tree
        switch (allowed >> 2) {
#define CH_C(N) case N: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * N) : "memory"); break;
            CH_C(1) CH_C(2) CH_C(3) CH_C(4) CH_C(5) CH_C(6) CH_C(7) CH_C(8) CH_C(9) CH_C(10) CH_C(11) CH_C(12) CH_C(13) CH_C(14) CH_C(15)
#undef CH_C
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        // All LDS reads but the youngest CH_RD have returned: the reads still in flight are the read-ahead into THIS tile's pieces (issued last, and LDS
        // reads return in order), everything older - the fragment reads of the slots that are about to be freed - is done before this wave signals.
        // Until round 6 this relied on the MFMAs that consume those fragments sitting in front of the sched_barrier above; instruction selection
        // does not promise that (vitblock.hip's boundary, profiles/r6e_*: a build that sank them below the barrier failed parity now and then).
        // Free in the designed state.
        asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(CH_RD) : "memory");
#ifdef LWDETR_CH_TIMING
        const unsigned long long tb = __builtin_amdgcn_s_memrealtime();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef LWDETR_CH_TIMING
        tt_wait += tb - ta; tt_bar += __builtin_amdgcn_s_memrealtime() - tb;
#endif
        lim = a + NSLOT; lim = lim < np ? lim : np;
        int q = a + n + n_next + 1; q = q < np - 1 ? q : np - 1;       // last piece the NEXT tile needs (issued already: NSLOT >= 2 n + n_next + 2)
        need = *(__attribute__((address_space(3))) const int*)(uintptr_t)(seq_lds + 4 * ((unsigned)q % NSLOT));
        flush();                 // refill the freed slots (handing the DMAs out between the MFMAs instead measured slower: each issue
                                 // stalls the instruction stream ~110 cycles wherever it sits)
        __builtin_amdgcn_sched_barrier(0);
    }
};

struct EncParams {
    const void* in; long ld_in;           // PF: the C2f concat (M, ld_in), K5 channels; else `memory` rows (M, D)
    void* memory;                         // (B * S, D): written when PF
    void* om; void* cls; long ld_cls; float* cls_max;
    void* values[6]; int nl;
    const unsigned char* rowvalid; const unsigned char* notpad;
    const void* wstream; const float* vec; int np;
    long M;                               // input rows of this launch
    int npix, S, lsi;                     // input row m -> image b = m / npix, memory row b * S + lsi + m % npix
    int ncls;
    float eps_p, eps_e;
    unsigned mem_bytes, cls_bytes, in_bytes;
};

// One wave's 32 rows through: [PF: cv2 (K5 -> D) + SiLU + LayerNorm2d -> memory] -> value projections of all decoder layers ->
// enc_output Linear + LayerNorm -> output_memory -> class Linear -> class logits + their row maximum.
template <typename T, int D, bool PF, int K5>
__global__ __launch_bounds__(256, 1) void enc_chain_kernel(const EncParams p) {
    typedef typename Vec<T>::v8 V8;
    constexpr int KS = D / 16, NTI = D / 32, PPT = D / 64;
    constexpr int KSI = PF ? K5 / 16 : KS, PPT5 = K5 / 64;
    constexpr int NSLOT = 32;
    constexpr int NCT = 3;                          // class tiles (ncls <= 96)
    constexpr int VEC_F = (PF ? 3 * D : 0) + 3 * D + 32 * NCT + 6 * D;
    constexpr int VEC_B = (VEC_F * 4 + 4095) / 4096 * 4096, VEC_DPW = VEC_B / 4096;
    constexpr bool TWO_CHAINS = D == 256 && !PF;    // second accumulator chain per tile where the registers allow it (hipcc spills otherwise)
    static_assert(KS % CH_RD == 0 && KSI % CH_RD == 0, "the fragment read-ahead ring must divide every tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float* vec = (const float*)(smem + NSLOT * CH_PIECE_B);
    const float* b2s = vec; const float* gps = vec + D; const float* bps = vec + 2 * D;
    const float* bes = vec + (PF ? 3 * D : 0); const float* ges = bes + D; const float* bts = ges + D;
    const float* bcs = bts + D; const float* bvs = bcs + 32 * NCT;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;

    const long t0 = ((long)blockIdx.x * 4 + wave) * 32;
    const long mrow = t0 + j;
    const bool live = mrow < p.M;
    CH_TS(0);
    // memory-space row of this lane's token
    const long img = live ? mrow / p.npix : 0;
    const long mm = live ? img * p.S + p.lsi + (mrow - img * p.npix) : 0;

    // ---- input rows first (they are older than every weight DMA: nothing queues behind the ring fill), then vectors + ring
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
    const unsigned in_off = live ? (unsigned)(mrow * p.ld_in * 2) : 0x80000000u;
    V8 xin[KSI];
#pragma unroll
    for (int t = 0; t < KSI; ++t)
        xin[t] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(r_in, in_off + (unsigned)((16 * t + 8 * h) * 2), 0, 0));
    const unsigned char rv = live ? p.rowvalid[mm] : 0, npd = live ? p.notpad[mm] : 0;

    WRing<NSLOT> ring;
    ring.src = (const char*)p.wstream; ring.lds0 = lds0; ring.np = p.np; ring.issued = 0; ring.wave = wave; ring.lane16 = lane16;
    {
        const char* vsrc = (const char*)p.vec;
#pragma unroll
        for (int i = 0; i < VEC_DPW; ++i) {
            const unsigned kb = (unsigned)(wave * VEC_DPW + i) * 1024u;
            ring.dma1k(vsrc, kb + lane16, lds0 + NSLOT * CH_PIECE_B + kb);
        }
        // only the pieces of the first tile (+ read-ahead + a few) now: the first begin_tile tops the ring up. The compiler waits for
        // the input rows with a vmcnt that does not know of the DMAs: every DMA issued before that wait would have to land first.
        ring.fill(2 * (PF ? PPT5 : PPT) + 2);            // (begin_tile_x: the first tile fetches the wait count of the second)
    }
    // an explicit vmcnt(0) the compiler can see (a real S_WAITCNT, not inline assembly): hipcc's own waits for the input rows and flags
    // are satisfied HERE, before the ring fill of the first begin_tile, and it adds none behind it
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ring.seq_init(lds0 + NSLOT * CH_PIECE_B + VEC_B + (unsigned)wave * NSLOT * 4);
    CH_TS(1);
    auto frag = [&](int g) -> V8 {           // global fragment index g = 4 * piece + fragment
        return *(const V8*)(smem + (((unsigned)g & (NSLOT * 4 - 1)) << 10) + lane16);
    };
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
    const __amdgpu_buffer_rsrc_t r_mem = __builtin_amdgcn_make_buffer_rsrc(p.memory ? p.memory : p.om, 0, p.memory ? (int)p.mem_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_om = __builtin_amdgcn_make_buffer_rsrc(p.om, 0, (int)p.mem_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_cls = __builtin_amdgcn_make_buffer_rsrc(p.cls, 0, (int)p.cls_bytes, 0x00020000);
    const unsigned row_off = live ? (unsigned)(mm * D * 2) : 0x80000000u;           // byte offset of the row in (B * S, D) tensors
    const unsigned cls_off = live ? (unsigned)(mm * p.ld_cls * 2) : 0x80000000u;

    int pc = 0;                                  // next piece
    V8 fr[CH_RD];
#ifdef LWDETR_CH_TIMING
    unsigned long long tt_mfma = 0, tt_epi = 0;
#endif
    // 32 channels x 32 rows: nf fragments against x[0 .. nf), fragment stream position 4 * pc
    auto tile = [&](auto& x, auto nf_tag, f32x16 acc) -> f32x16 {
        constexpr int nf = decltype(nf_tag)::value;
        const int g0 = 4 * pc;
        // two independent accumulator chains (even / odd k-steps): a 32x32x16 MFMA that accumulates into the result of the one right
        // in front of it waits for that result
        f32x16 acc2 = {};
#pragma unroll
        for (int f = 0; f < nf; ++f) {
            const V8 a = fr[f % CH_RD];
            fr[f % CH_RD] = frag(g0 + f + CH_RD);
            if (TWO_CHAINS && (f & 1)) acc2 = Mma32c<T>::k16(a, x[f], acc2);
            else acc = Mma32c<T>::k16(a, x[f], acc);
            // pin the order: hipcc otherwise sinks every fragment read to just in front of its MFMA (each MFMA then waits a whole LDS
            // round trip: 114 - 139 cycles per MFMA slot measured, profiles/r4c_enc_chain_phase_timing.txt)
            __builtin_amdgcn_sched_barrier(0);
        }
        pc += nf / 4;
        if (TWO_CHAINS) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
        }
        return acc;
    };
    // rows rounded to T held as packed pairs (dword d of tile n = registers 2 d, 2 d + 1): LayerNorm with affine, stores the
    // result to `rs` rows (16-byte pieces) and leaves it as B operands in xo (k-slot order of an accumulator hand-over)
    auto layernorm_store = [&](unsigned (&xp)[NTI][8], float s, const float* gam, const float* bet, float eps,
                               const __amdgpu_buffer_rsrc_t& rs, V8 (&xo)[KS]) {
        s += __shfl_xor(s, 32);
        const float mean = s * (1.f / D);
        float v = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n)
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                float v0, v1; cunpack2<T>(xp[n][d], v0, v1);
                v0 -= mean; v1 -= mean;
                v = fmaf(v0, v0, v); v = fmaf(v1, v1, v);
            }
        v += __shfl_xor(v, 32);
        const float rstd = 1.f / sqrtf(v * (1.f / D) + eps);
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            unsigned w[8];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c0 = 32 * n + 8 * b + 4 * h;
                const f32x4 g = *(const f32x4*)(gam + c0), be = *(const f32x4*)(bet + c0);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    float v0, v1; cunpack2<T>(xp[n][2 * b + d], v0, v1);
                    w[2 * b + d] = cpack2<T>(fmaf((v0 - mean) * rstd, g[2 * d], be[2 * d]), fmaf((v1 - mean) * rstd, g[2 * d + 1], be[2 * d + 1]));
                }
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const cu32x4 ow = crows8(w[4 * jb], w[4 * jb + 1], w[4 * jb + 2], w[4 * jb + 3]);
                __builtin_amdgcn_raw_buffer_store_b128(ow, rs, row_off + (unsigned)((32 * n + 16 * jb + 8 * h) * 2), 0, 0);
            }
            xo[2 * n] = __builtin_bit_cast(V8, cu32x4{w[0], w[1], w[2], w[3]});
            xo[2 * n + 1] = __builtin_bit_cast(V8, cu32x4{w[4], w[5], w[6], w[7]});
        }
        ring.stores(2 * NTI);
    };

    V8 xf[KS];                                   // `memory` rows as B operands
    // first tile: its pieces (and the two read ahead) have landed; the read-ahead ring starts
    ring.begin_tile_x(0, PF ? PPT5 : PPT, PF ? PPT5 : PPT);
#pragma unroll
    for (int i = 0; i < CH_RD; ++i) fr[i] = frag(i);
    CH_TS(2);
    if constexpr (PF) {
        // ---- projector: C2f.cv2 (1x1 conv, BatchNorm folded) + SiLU, LayerNorm over channels -> memory (projector.py:117-132)
        unsigned xp[NTI][8];
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            if (n > 0) ring.begin_tile_x(pc, PPT5, n + 1 < NTI ? PPT5 : PPT);
            const f32x16 acc = tile(xin, std::integral_constant<int, KSI>{}, bias16(b2s + 32 * n));
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const float a0 = acc[2 * d], a1 = acc[2 * d + 1];
                const unsigned w = cpack2<T>(a0 / (1.f + __expf(-a0)), a1 / (1.f + __expf(-a1)));
                xp[n][d] = w;
                float v0, v1; cunpack2<T>(w, v0, v1);
                s += v0 + v1;
            }
        }
        layernorm_store(xp, s, gps, bps, p.eps_p, r_mem, xf);
    } else {
#pragma unroll
        for (int t = 0; t < KS; ++t) xf[t] = xin[t];
    }
    CH_TS(3);
    // ---- value projections of all decoder layers (ms_deform_attn.py:110-114: masked_fill of the OUTPUT rows of padded pixels)
    {
        const int nvt = p.nl * NTI;
#pragma unroll 1
        for (int vt = 0; vt < nvt; ++vt) {
            if (PF || vt > 0) ring.begin_tile_x(pc, PPT, PPT);
#ifdef LWDETR_CH_TIMING
            const unsigned long long tv0 = __builtin_amdgcn_s_memrealtime();
#endif
            f32x16 acc = tile(xf, std::integral_constant<int, KS>{}, bias16(bvs + 32 * vt));
#ifdef LWDETR_CH_TIMING
            asm volatile("" : "+v"(acc));
            const unsigned long long tv1 = __builtin_amdgcn_s_memrealtime();
#endif
            const int li = vt / NTI, n = vt - li * NTI;
            const __amdgpu_buffer_rsrc_t r_v = __builtin_amdgcn_make_buffer_rsrc(p.values[li], 0, (int)p.mem_bytes, 0x00020000);
            if (!npd) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const cu32x4 ow = crows8(cpack2<T>(acc[8 * jb], acc[8 * jb + 1]), cpack2<T>(acc[8 * jb + 2], acc[8 * jb + 3]),
                                         cpack2<T>(acc[8 * jb + 4], acc[8 * jb + 5]), cpack2<T>(acc[8 * jb + 6], acc[8 * jb + 7]));
                __builtin_amdgcn_raw_buffer_store_b128(ow, r_v, row_off + (unsigned)((32 * n + 16 * jb + 8 * h) * 2), 0, 0);
            }
            ring.stores(2);
#ifdef LWDETR_CH_TIMING
            tt_mfma += tv1 - tv0; tt_epi += __builtin_amdgcn_s_memrealtime() - tv1;
#endif
        }
    }
    CH_TS(4);
    // ---- enc_output Linear on the rows (invalid proposals: the INPUT row is zeroed, transformer.py:113-116) + LayerNorm -> output_memory
    {
        if (!rv) {
#pragma unroll
            for (int t = 0; t < KS; ++t) xf[t] = V8{};
        }
        unsigned xp[NTI][8];
        float s = 0.f;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            ring.begin_tile_x(pc, PPT, PPT);
            const f32x16 acc = tile(xf, std::integral_constant<int, KS>{}, bias16(bes + 32 * n));
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const unsigned w = cpack2<T>(acc[2 * d], acc[2 * d + 1]);
                xp[n][d] = w;
                float v0, v1; cunpack2<T>(w, v0, v1);
                s += v0 + v1;
            }
        }
        layernorm_store(xp, s, ges, bts, p.eps_e, r_om, xf);
    }
    CH_TS(5);
    // ---- class logits of every token and their maximum (the two-stage selection score, transformer.py:244-246)
    {
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < NCT; ++n) {
            ring.begin_tile_x(pc, PPT, PPT);
            const f32x16 acc = tile(xf, std::integral_constant<int, KS>{}, bias16(bcs + 32 * n));
            unsigned w[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                w[d] = cpack2<T>(acc[2 * d], acc[2 * d + 1]);
                float v0, v1; cunpack2<T>(w[d], v0, v1);
                const int c = 32 * n + 8 * (d >> 1) + 4 * h + 2 * (d & 1);
                if (c < p.ncls) mx = fmaxf(mx, v0);
                if (c + 1 < p.ncls) mx = fmaxf(mx, v1);
            }
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const cu32x4 ow = crows8(w[4 * jb], w[4 * jb + 1], w[4 * jb + 2], w[4 * jb + 3]);
                __builtin_amdgcn_raw_buffer_store_b128(ow, r_cls, cls_off + (unsigned)((32 * n + 16 * jb + 8 * h) * 2), 0, 0);
            }
            ring.stores(2);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (live && h == 0) p.cls_max[mm] = mx;
    }
    CH_TS(6);
#ifdef LWDETR_CH_TIMING
    if ((blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && lane == 0) { g_ch_timing[blockIdx.x != 0][wave][13] = ring.tt_wait; g_ch_timing[blockIdx.x != 0][wave][14] = ring.tt_bar;
        g_ch_timing[blockIdx.x != 0][wave][11] = tt_mfma; g_ch_timing[blockIdx.x != 0][wave][12] = tt_epi; }
#endif
}

// ------------------------------------------------------------------------------------------------ generic row chain
// A run-time program of up to 6 stages over a wave's 32 rows (decoder: output projections + LayerNorm (+ the sampling-offset /
// attention-weight Linear), the box / class heads, the two-stage box head, ref_point_head):
//   FULL stage: y = W x + b (+ residual rows loaded in the prologue) (ReLU) -> rounded -> (LayerNorm with affine -> rounded)
//               (-> stored to `out` rows) -> becomes the operand of the following stages (optionally + qpos, rounded: the
//               reference's `tgt + query_pos`); N = D. The first FULL stage of a chain may contract over K0 = 2 D input channels.
//   SIDE stage: out[:, 0:ncols] = W x + b from the CURRENT operand, tile by tile; the operand stays.
enum { CS_FULL = 0, CS_SIDE = 1 };
enum { CF_RES = 1, CF_RELU = 2, CF_LN = 4, CF_STORE = 8, CF_ADDQ = 16 };
struct ChainStageK { int kind, nt, flags, ncols, bias_off, gam_off, bet_off; float eps; void* out; long ldo; unsigned out_bytes; };
struct MlpChainParams {
    const void* in; long ld_in; unsigned in_bytes;
    const void* res; long ld_res; unsigned res_bytes;
    const void* qpos; long ld_q; unsigned q_bytes;
    const void* wstream; const float* vec; int vec_dpw; int np; long M; int nst;
    ChainStageK st[6];
};
constexpr int CH_VEC_MAX_B = 16384;

template <typename T, int D, int KS0, bool RES, bool QP>
__global__ __launch_bounds__(256, 1) void mlp_chain_kernel(const MlpChainParams p) {
    typedef typename Vec<T>::v8 V8;
    constexpr int KS = D / 16, NTI = D / 32, PPT = D / 64, PPT0 = KS0 / 4;
    constexpr int NSLOT = 32;
    constexpr bool TWO_CHAINS = false;              // (registers: the rolled stage loop of this form is at the limit already)
    static_assert(KS % CH_RD == 0 && KS0 % CH_RD == 0, "the fragment read-ahead ring must divide every tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float* vec = (const float*)(smem + NSLOT * CH_PIECE_B);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const long t0 = ((long)blockIdx.x * 4 + wave) * 32;
    const long mrow = t0 + j;
    const bool live = mrow < p.M;

    // ---- prologue loads (older than every weight DMA): input rows as B operands (natural k order), residual / qpos rows in
    // accumulator layout (16-byte pieces: channels 32 n + 16 jb + 8 h .. + 7 of row j)
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
    const unsigned in_off = live ? (unsigned)(mrow * p.ld_in * 2) : 0x80000000u;
    V8 xin[KS0];
#pragma unroll
    for (int t = 0; t < KS0; ++t)
        xin[t] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(r_in, in_off + (unsigned)((16 * t + 8 * h) * 2), 0, 0));
    cu32x4 xr[RES ? NTI : 1][2], xq[QP ? NTI : 1][2];
    if constexpr (RES) {
        const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)p.res_bytes, 0x00020000);
        const unsigned off = live ? (unsigned)(mrow * p.ld_res * 2) : 0x80000000u;
#pragma unroll
        for (int n = 0; n < NTI; ++n)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) xr[n][jb] = __builtin_amdgcn_raw_buffer_load_b128(r_res, off + (unsigned)((32 * n + 16 * jb + 8 * h) * 2), 0, 0);
    }
    if constexpr (QP) {
        const __amdgpu_buffer_rsrc_t r_q = __builtin_amdgcn_make_buffer_rsrc((void*)p.qpos, 0, (int)p.q_bytes, 0x00020000);
        const unsigned off = live ? (unsigned)(mrow * p.ld_q * 2) : 0x80000000u;
#pragma unroll
        for (int n = 0; n < NTI; ++n)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) xq[n][jb] = __builtin_amdgcn_raw_buffer_load_b128(r_q, off + (unsigned)((32 * n + 16 * jb + 8 * h) * 2), 0, 0);
    }
    WRing<NSLOT> ring;
    ring.src = (const char*)p.wstream; ring.lds0 = lds0; ring.np = p.np; ring.issued = 0; ring.wave = wave; ring.lane16 = lane16;
    {
        const char* vsrc = (const char*)p.vec;
        for (int i = 0; i < p.vec_dpw; ++i) {
            const unsigned kb = (unsigned)(wave * p.vec_dpw + i) * 1024u;
            ring.dma1k(vsrc, kb + lane16, lds0 + NSLOT * CH_PIECE_B + kb);
        }
        ring.fill(PPT0 + 2 + 4);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0), visible to the compiler: see enc_chain_kernel
    auto frag = [&](int g) -> V8 { return *(const V8*)(smem + (((unsigned)g & (NSLOT * 4 - 1)) << 10) + lane16); };
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
    int pc = 0;
    bool first = true;
    V8 fr[CH_RD];
    auto tile = [&](auto& x, auto nf_tag, f32x16 acc) -> f32x16 {
        constexpr int nf = decltype(nf_tag)::value;
        const int g0 = 4 * pc;
        // two independent accumulator chains (even / odd k-steps): a 32x32x16 MFMA that accumulates into the result of the one right
        // in front of it waits for that result
        f32x16 acc2 = {};
#pragma unroll
        for (int f = 0; f < nf; ++f) {
            const V8 a = fr[f % CH_RD];
            fr[f % CH_RD] = frag(g0 + f + CH_RD);
            if (TWO_CHAINS && (f & 1)) acc2 = Mma32c<T>::k16(a, x[f], acc2);
            else acc = Mma32c<T>::k16(a, x[f], acc);
            // pin the order: hipcc otherwise sinks every fragment read to just in front of its MFMA (each MFMA then waits a whole LDS
            // round trip: 114 - 139 cycles per MFMA slot measured, profiles/r4c_enc_chain_phase_timing.txt)
            __builtin_amdgcn_sched_barrier(0);
        }
        pc += nf / 4;
        if (TWO_CHAINS) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
        }
        return acc;
    };
    auto next_tile = [&](int n) {
        if (first) first = false;
        else ring.template begin_tile<NSLOT - 2 * PPT - 2>(pc, n);
    };
    V8 xf[KS];
    if constexpr (KS0 == KS) {
#pragma unroll
        for (int t = 0; t < KS; ++t) xf[t] = xin[t];
    }
    ring.template begin_tile<4>(0, PPT0);
#pragma unroll
    for (int i = 0; i < CH_RD; ++i) fr[i] = frag(i);

    // FULL stage over operand x (nf fragments per tile)
    auto full = [&](const ChainStageK& st, auto& x, auto nf_tag) {
        constexpr int nf = decltype(nf_tag)::value;
        const float* bsrc = vec + st.bias_off;
        unsigned xp[NTI][8];
        float s = 0.f;
        const bool relu = st.flags & CF_RELU, has_res = RES && (st.flags & CF_RES);
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            next_tile(nf / 4);
            f32x16 init = bias16(bsrc + 32 * n);
            if constexpr (RES) {
                if (has_res) {
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const cu32x4 own = crows8(xr[n][jb][0], xr[n][jb][1], xr[n][jb][2], xr[n][jb][3]);     // its own inverse
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned ow_ = own[q];
                            float v0, v1; cunpack2<T>(ow_, v0, v1);
                            init[8 * jb + 2 * q] += v0; init[8 * jb + 2 * q + 1] += v1;
                        }
                    }
                }
            }
            const f32x16 acc = tile(x, nf_tag, init);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                float a0 = acc[2 * d], a1 = acc[2 * d + 1];
                if (relu) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); }
                const unsigned w = cpack2<T>(a0, a1);
                xp[n][d] = w;
                float v0, v1; cunpack2<T>(w, v0, v1);
                s += v0 + v1;
            }
        }
        float mean = 0.f, rstd = 1.f;
        const bool ln = st.flags & CF_LN;
        if (ln) {
            s += __shfl_xor(s, 32);
            mean = s * (1.f / D);
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < NTI; ++n)
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    float v0, v1; cunpack2<T>(xp[n][d], v0, v1);
                    v0 -= mean; v1 -= mean;
                    v = fmaf(v0, v0, v); v = fmaf(v1, v1, v);
                }
            v += __shfl_xor(v, 32);
            rstd = 1.f / sqrtf(v * (1.f / D) + st.eps);
        }
        const float* gam = vec + st.gam_off; const float* bet = vec + st.bet_off;
        const bool store = st.flags & CF_STORE, addq = QP && (st.flags & CF_ADDQ);
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(store ? st.out : (void*)p.in, 0, store ? (int)st.out_bytes : 0, 0x00020000);
        const unsigned ooff = live ? (unsigned)(mrow * st.ldo * 2) : 0x80000000u;
#pragma unroll
        for (int n = 0; n < NTI; ++n) {
            unsigned w[8];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c0 = 32 * n + 8 * b + 4 * h;
                if (ln) {
                    const f32x4 g = *(const f32x4*)(gam + c0), be = *(const f32x4*)(bet + c0);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        float v0, v1; cunpack2<T>(xp[n][2 * b + d], v0, v1);
                        w[2 * b + d] = cpack2<T>(fmaf((v0 - mean) * rstd, g[2 * d], be[2 * d]), fmaf((v1 - mean) * rstd, g[2 * d + 1], be[2 * d + 1]));
                    }
                } else {
                    w[2 * b] = xp[n][2 * b]; w[2 * b + 1] = xp[n][2 * b + 1];
                }
            }
            if (store) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const cu32x4 ow = crows8(w[4 * jb], w[4 * jb + 1], w[4 * jb + 2], w[4 * jb + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(ow, r_out, ooff + (unsigned)((32 * n + 16 * jb + 8 * h) * 2), 0, 0);
                }
            }
            if constexpr (QP) {
                if (addq) {              // operand of the following stages = T(x + qpos): the reference's `tgt + query_pos`
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const cu32x4 own = crows8(xq[n][jb][0], xq[n][jb][1], xq[n][jb][2], xq[n][jb][3]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const unsigned qw = own[q];
                            float q0, q1, v0, v1; cunpack2<T>(qw, q0, q1); cunpack2<T>(w[4 * jb + q], v0, v1);
                            w[4 * jb + q] = cpack2<T>(v0 + q0, v1 + q1);
                        }
                    }
                }
            }
            xf[2 * n] = __builtin_bit_cast(V8, cu32x4{w[0], w[1], w[2], w[3]});
            xf[2 * n + 1] = __builtin_bit_cast(V8, cu32x4{w[4], w[5], w[6], w[7]});
        }
    };
    // SIDE stage from the current operand
    auto side = [&](const ChainStageK& st) {
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(st.out, 0, (int)st.out_bytes, 0x00020000);
        const unsigned ooff = live ? (unsigned)(mrow * st.ldo * 2) : 0x80000000u;
        const bool wide = (st.ldo % 8 == 0) && (st.ncols % 8 == 0);
        const float* bsrc = vec + st.bias_off;
#pragma unroll 1
        for (int vt = 0; vt < st.nt; ++vt) {
            next_tile(PPT);
            const f32x16 acc = tile(xf, std::integral_constant<int, KS>{}, bias16(bsrc + 32 * vt));
            unsigned w[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) w[d] = cpack2<T>(acc[2 * d], acc[2 * d + 1]);
            if (wide) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const cu32x4 ow = crows8(w[4 * jb], w[4 * jb + 1], w[4 * jb + 2], w[4 * jb + 3]);
                    const int c0 = 32 * vt + 16 * jb + 8 * h;
                    __builtin_amdgcn_raw_buffer_store_b128(ow, r_out, c0 < st.ncols ? ooff + (unsigned)(c0 * 2) : 0x80000000u, 0, 0);
                }
            } else {                // 8-byte pieces: channels 32 vt + 8 b + 4 h .. + 3 (ldo % 4 == 0; pad columns up to ceil4(ncols) are written)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int c0 = 32 * vt + 8 * b + 4 * h;
                    typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(cu32x2{w[2 * b], w[2 * b + 1]}, r_out, c0 < st.ncols ? ooff + (unsigned)(c0 * 2) : 0x80000000u, 0, 0);
                }
            }
        }
    };
#pragma unroll 1
    for (int si = 0; si < p.nst; ++si) {
        const ChainStageK& st = p.st[si];
        if (st.kind == CS_SIDE) side(st);
        else if (KS0 != KS && si == 0) full(st, xin, std::integral_constant<int, KS0>{});
        else full(st, xf, std::integral_constant<int, KS>{});
    }
}

template <typename T, int D, bool PF, int K5>
int launch_enc(const EncParams& p, hipStream_t st) {
    constexpr int NSLOT = 32, NCT = 3;
    constexpr int VEC_F = (PF ? 3 * D : 0) + 3 * D + 32 * NCT + 6 * D;
    constexpr int VEC_B = (VEC_F * 4 + 4095) / 4096 * 4096;
    constexpr size_t lds = (size_t)NSLOT * CH_PIECE_B + VEC_B + 4 * NSLOT * 4;      // ring, vectors, sequence numbers of the 4 waves
    static bool attr_done[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)enc_chain_kernel<T, D, PF, K5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    const long grid = (p.M + 127) / 128;
    const double kin = PF ? (double)K5 : 0.0;
    ProfScope ps(KID_CHAIN, 2.0 * p.M * D * (kin + D + 32.0 * NCT + (double)p.nl * D),
                 (double)p.M * 2.0 * ((PF ? K5 + D : D) + D + 96.0 + (double)p.nl * D) + 4.0 * p.M, st);
    hipLaunchKernelGGL((enc_chain_kernel<T, D, PF, K5>), dim3((unsigned)grid), dim3(256), lds, st, p);
    return lwdetr_check_launch();
}

template <typename T>
int dispatch_enc(const EncParams& p, int D, int k5, hipStream_t st) {
    if (D == 256 && k5 == 640) return launch_enc<T, 256, true, 640>(p, st);
    if (D == 256 && k5 == 0) return launch_enc<T, 256, false, 640>(p, st);
    if (D == 384 && k5 == 0) return launch_enc<T, 384, false, 960>(p, st);
    return LWDETR_ERR_UNSUPPORTED;
}

// ---- the same program, output channels split over the 4 waves of a workgroup that owns 32 rows (few rows: decoder queries).
// The row-per-wave form above runs all N / 32 tiles of a stage one after the other on one wave: 128 dependent MFMAs per 256 x 256
// stage, 10 - 25 us per chain whatever the row count (profiles/r4b_*), slower than the launches it replaces when the rows do not fill
// the chip. Here wave w computes tiles w, w + 4, ... of every stage; the operand of a stage lives in LDS as B fragments (every wave
// reads all of it into registers at the start of the stage), LayerNorm statistics go through LDS (two passes, as the row kernel),
// and the result is written back as the fragments (2 n, 2 n + 1) of tile n - the accumulator hand-over of the row-per-wave form,
// through LDS. Same packed weight stream, same vectors, same rounding points.
template <typename T, int D>
__global__ __launch_bounds__(256, 1) void mlp_chain_split_kernel(const MlpChainParams p) {
    typedef typename Vec<T>::v8 V8;
    constexpr int KS = D / 16, NTI = D / 32, PPT = D / 64, TPW = NTI / 4;     // tiles per wave and FULL stage
    // A step = the pieces of up to 4 tiles (one per wave) that land together. D = 256: a tile's 4 pieces; D = 384: HALF a tile's 6 pieces
    // (the stream is packed k-half-major inside a group of 4 tiles, lwdetr_amd/kernels.py:chain_pieces_split): two steps fit the ring
    constexpr int HALVES = D == 384 ? 2 : 1, PPH = PPT / HALVES, KSH = KS / HALVES;
    constexpr int NSLOT = D == 384 ? 24 : 32;
    static_assert(NTI % 4 == 0 && PPT % HALVES == 0, "tiles split evenly over 4 waves");
    static_assert(2 * 4 * PPH <= NSLOT, "the ring holds two steps (a step's pieces are issued one step ahead)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int vec_b = p.vec_dpw * 4096;
    const float* vec = (const float*)(smem + NSLOT * CH_PIECE_B);
    char* act = smem + NSLOT * CH_PIECE_B + vec_b;                              // KS fragments of 1 KB: the current operand
    float* stats = (float*)(act + KS * 1024);                                   // [4 waves][32 rows]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const long mrow = (long)blockIdx.x * 32 + j;
    const bool live = mrow < p.M;

    // ---- prologue: input rows -> B fragments t = wave, wave + 4, ... in LDS; residual / qpos rows of this wave's tiles in registers
    const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
    const unsigned in_off = live ? (unsigned)(mrow * p.ld_in * 2) : 0x80000000u;
    cu32x4 xi[KS / 4];
#pragma unroll
    for (int i = 0; i < KS / 4; ++i) xi[i] = __builtin_amdgcn_raw_buffer_load_b128(r_in, in_off + (unsigned)((16 * (wave + 4 * i) + 8 * h) * 2), 0, 0);
    cu32x4 xr[TPW][2], xq[TPW][2];
    const bool has_r = p.res != nullptr, has_q = p.qpos != nullptr;
    {
        const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(has_r ? (void*)p.res : (void*)p.in, 0, has_r ? (int)p.res_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_q = __builtin_amdgcn_make_buffer_rsrc(has_q ? (void*)p.qpos : (void*)p.in, 0, has_q ? (int)p.q_bytes : 0, 0x00020000);
        const unsigned offr = live ? (unsigned)(mrow * p.ld_res * 2) : 0x80000000u, offq = live ? (unsigned)(mrow * p.ld_q * 2) : 0x80000000u;
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const unsigned c = (unsigned)((32 * (wave + 4 * i) + 16 * jb + 8 * h) * 2);
                xr[i][jb] = __builtin_amdgcn_raw_buffer_load_b128(r_res, offr + c, 0, 0);       // out of range (no residual): zeros
                xq[i][jb] = __builtin_amdgcn_raw_buffer_load_b128(r_q, offq + c, 0, 0);
            }
    }
    WRing<NSLOT> ring;
    ring.src = (const char*)p.wstream; ring.lds0 = lds0; ring.np = p.np; ring.issued = 0; ring.wave = wave; ring.lane16 = lane16;
    {
        const char* vsrc = (const char*)p.vec;
        for (int i = 0; i < p.vec_dpw; ++i) {
            const unsigned kb = (unsigned)(wave * p.vec_dpw + i) * 1024u;
            ring.dma1k(vsrc, kb + lane16, lds0 + NSLOT * CH_PIECE_B + kb);
        }
        ring.fill(NSLOT);
    }
#pragma unroll
    for (int i = 0; i < KS / 4; ++i) *(cu32x4*)(act + (wave + 4 * i) * 1024 + lane16) = xi[i];
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): residual / qpos rows are in registers before the stage loop (no waits in it)

    auto frag = [&](int g) -> V8 { return *(const V8*)(smem + (((unsigned)g % (NSLOT * 4)) << 10) + lane16); };
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
    // group of up to 4 tiles (one per wave) starting at piece a, n pieces: all of them have landed, everybody is done with the pieces
    // below a (and with the operand reads / writes before this point), the ring is topped up
    auto begin_group = [&](int a, int n) {
        __builtin_amdgcn_sched_barrier(0);
        int b = a + n; b = b < p.np ? b : p.np;
        const int v = ring.issued - b;
        if (v <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else ch_wait_le(v);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int lim = a + NSLOT; lim = lim < p.np ? lim : p.np;
        while (ring.issued < lim) { ring.dma_piece(ring.issued); ++ring.issued; }
        __builtin_amdgcn_sched_barrier(0);
    };
    // one step of a 32-channel tile: KSH fragments from piece `pc0` against the operand registers x[x0 .. x0 + KSH)
    auto tile = [&](const V8 (&x)[KS], auto x0_tag, int pc0, f32x16 acc) -> f32x16 {
        constexpr int x0 = decltype(x0_tag)::value;
        const int g0 = 4 * pc0;
        V8 fr[CH_RD];
#pragma unroll
        for (int i = 0; i < CH_RD; ++i) fr[i] = frag(g0 + i);
        f32x16 acc2 = {};                       // two independent accumulator chains (even / odd k-steps)
#pragma unroll
        for (int f = 0; f < KSH; ++f) {
            const V8 a = fr[f % CH_RD];
            if (f + CH_RD < KSH) fr[f % CH_RD] = frag(g0 + f + CH_RD);
            if (f & 1) acc2 = Mma32c<T>::k16(a, x[x0 + f], acc2);
            else acc = Mma32c<T>::k16(a, x[x0 + f], acc);
            __builtin_amdgcn_sched_barrier(0);      // keep the fragment reads CH_RD MFMAs ahead (see enc_chain_kernel)
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += acc2[e];
        return acc;
    };
    auto add_own = [&](f32x16& init, const cu32x4 (&rows)[2]) {      // rows in accumulator layout (16-byte pieces) -> this lane's registers
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const cu32x4 own = crows8(rows[jb][0], rows[jb][1], rows[jb][2], rows[jb][3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned ow_ = own[q];
                float v0, v1; cunpack2<T>(ow_, v0, v1);
                init[8 * jb + 2 * q] += v0; init[8 * jb + 2 * q + 1] += v1;
            }
        }
    };
    int pc = 0;
#pragma unroll 1
    for (int si = 0; si < p.nst; ++si) {
        // static indices + scalar selects: a run-time index (or a struct assigned in a switch) makes hipcc keep the stage table in scratch
#define CH_STF(f) (si == 0 ? p.st[0].f : si == 1 ? p.st[1].f : si == 2 ? p.st[2].f : si == 3 ? p.st[3].f : si == 4 ? p.st[4].f : p.st[5].f)
        struct { int kind, nt, flags, ncols, bias_off, gam_off, bet_off; float eps; void* out; long ldo; unsigned out_bytes; } st;
        st.kind = CH_STF(kind); st.nt = CH_STF(nt); st.flags = CH_STF(flags); st.ncols = CH_STF(ncols); st.bias_off = CH_STF(bias_off);
        st.gam_off = CH_STF(gam_off); st.bet_off = CH_STF(bet_off); st.eps = CH_STF(eps); st.out = CH_STF(out); st.ldo = CH_STF(ldo);
        st.out_bytes = CH_STF(out_bytes);
#undef CH_STF
        const float* bsrc = vec + st.bias_off;
        if (st.kind == CS_SIDE) {
            const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(st.out, 0, (int)st.out_bytes, 0x00020000);
            const unsigned ooff = live ? (unsigned)(mrow * st.ldo * 2) : 0x80000000u;
            const bool wide = (st.ldo % 8 == 0) && (st.ncols % 8 == 0);
            V8 xf[KS];
            bool have = false;
#pragma unroll 1
            for (int g0t = 0; g0t < st.nt; g0t += 4) {
                const int ng = st.nt - g0t < 4 ? st.nt - g0t : 4;
                const int vt = g0t + wave;
                f32x16 acc = bias16(bsrc + 32 * (vt < st.nt ? vt : 0));
                begin_group(pc, ng * PPH);
                if (!have) {
#pragma unroll
                    for (int t = 0; t < KS; ++t) xf[t] = *(const V8*)(act + t * 1024 + lane16);
                    have = true;
                }
                if (wave < ng) acc = tile(xf, std::integral_constant<int, 0>{}, pc + wave * PPH, acc);
                pc += ng * PPH;
                if constexpr (HALVES == 2) {
                    begin_group(pc, ng * PPH);
                    if (wave < ng) acc = tile(xf, std::integral_constant<int, KSH>{}, pc + wave * PPH, acc);
                    pc += ng * PPH;
                }
                if (wave < ng) {
                    unsigned w[8];
#pragma unroll
                    for (int d = 0; d < 8; ++d) w[d] = cpack2<T>(acc[2 * d], acc[2 * d + 1]);
                    if (wide) {
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb) {
                            const cu32x4 ow = crows8(w[4 * jb], w[4 * jb + 1], w[4 * jb + 2], w[4 * jb + 3]);
                            const int c0 = 32 * vt + 16 * jb + 8 * h;
                            __builtin_amdgcn_raw_buffer_store_b128(ow, r_out, c0 < st.ncols ? ooff + (unsigned)(c0 * 2) : 0x80000000u, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int c0 = 32 * vt + 8 * b + 4 * h;
                            typedef unsigned int cu32x2 __attribute__((ext_vector_type(2)));
                            __builtin_amdgcn_raw_buffer_store_b64(cu32x2{w[2 * b], w[2 * b + 1]}, r_out, c0 < st.ncols ? ooff + (unsigned)(c0 * 2) : 0x80000000u, 0, 0);
                        }
                    }
                }
            }
            continue;
        }
        // ---- FULL stage
        const bool relu = st.flags & CF_RELU, use_r = st.flags & CF_RES, ln = st.flags & CF_LN, store = st.flags & CF_STORE, addq = st.flags & CF_ADDQ;
        unsigned xp[TPW][8];
        float s = 0.f;
        {
            V8 xf[KS];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int n = wave + 4 * i;
                f32x16 acc = bias16(bsrc + 32 * n);
                if (use_r) add_own(acc, xr[i]);
                begin_group(pc, 4 * PPH);
                if (i == 0) {
#pragma unroll
                    for (int t = 0; t < KS; ++t) xf[t] = *(const V8*)(act + t * 1024 + lane16);
                }
                acc = tile(xf, std::integral_constant<int, 0>{}, pc + wave * PPH, acc);
                pc += 4 * PPH;
                if constexpr (HALVES == 2) {
                    begin_group(pc, 4 * PPH);
                    acc = tile(xf, std::integral_constant<int, KSH>{}, pc + wave * PPH, acc);
                    pc += 4 * PPH;
                }
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    float a0 = acc[2 * d], a1 = acc[2 * d + 1];
                    if (relu) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); }
                    const unsigned w = cpack2<T>(a0, a1);
                    xp[i][d] = w;
                    float v0, v1; cunpack2<T>(w, v0, v1);
                    s += v0 + v1;
                }
            }
        }
        float mean = 0.f, rstd = 1.f;
        if (ln) {       // two passes over the rounded row, partial sums of the 4 waves through LDS
            s += __shfl_xor(s, 32);
            if (h == 0) stats[wave * 32 + j] = s;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            mean = (stats[j] + stats[32 + j] + stats[64 + j] + stats[96 + j]) * (1.f / D);
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    float v0, v1; cunpack2<T>(xp[i][d], v0, v1);
                    v0 -= mean; v1 -= mean;
                    v = fmaf(v0, v0, v); v = fmaf(v1, v1, v);
                }
            v += __shfl_xor(v, 32);
            if (h == 0) stats[128 + wave * 32 + j] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            rstd = 1.f / sqrtf((stats[128 + j] + stats[160 + j] + stats[192 + j] + stats[224 + j]) * (1.f / D) + st.eps);
        }
        // every wave has read the operand into registers long ago (two group barriers back at least: TPW >= 2): overwrite it
        const float* gam = vec + st.gam_off; const float* bet = vec + st.bet_off;
        const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(store ? st.out : (void*)p.in, 0, store ? (int)st.out_bytes : 0, 0x00020000);
        const unsigned ooff = live ? (unsigned)(mrow * st.ldo * 2) : 0x80000000u;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int n = wave + 4 * i;
            unsigned w[8];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c0 = 32 * n + 8 * b + 4 * h;
                if (ln) {
                    const f32x4 g = *(const f32x4*)(gam + c0), be = *(const f32x4*)(bet + c0);
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        float v0, v1; cunpack2<T>(xp[i][2 * b + d], v0, v1);
                        w[2 * b + d] = cpack2<T>(fmaf((v0 - mean) * rstd, g[2 * d], be[2 * d]), fmaf((v1 - mean) * rstd, g[2 * d + 1], be[2 * d + 1]));
                    }
                } else {
                    w[2 * b] = xp[i][2 * b]; w[2 * b + 1] = xp[i][2 * b + 1];
                }
            }
            if (store) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const cu32x4 ow = crows8(w[4 * jb], w[4 * jb + 1], w[4 * jb + 2], w[4 * jb + 3]);
                    __builtin_amdgcn_raw_buffer_store_b128(ow, r_out, ooff + (unsigned)((32 * n + 16 * jb + 8 * h) * 2), 0, 0);
                }
            }
            if (addq) {
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const cu32x4 own = crows8(xq[i][jb][0], xq[i][jb][1], xq[i][jb][2], xq[i][jb][3]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned qw = own[q];
                        float q0, q1, v0, v1; cunpack2<T>(qw, q0, q1); cunpack2<T>(w[4 * jb + q], v0, v1);
                        w[4 * jb + q] = cpack2<T>(v0 + q0, v1 + q1);
                    }
                }
            }
            *(cu32x4*)(act + (2 * n) * 1024 + lane16) = cu32x4{w[0], w[1], w[2], w[3]};
            *(cu32x4*)(act + (2 * n + 1) * 1024 + lane16) = cu32x4{w[4], w[5], w[6], w[7]};
        }
    }
}

template <typename T, int D>
int launch_mlp_chain_split(const MlpChainParams& p, hipStream_t st, double flops, double bytes) {
    const size_t lds = (size_t)(D == 384 ? 24 : 32) * CH_PIECE_B + (size_t)p.vec_dpw * 4096 + (D / 16) * 1024 + 1024;
    if (lds > 160 * 1024) return LWDETR_ERR_UNSUPPORTED;
    static size_t attr_done[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    if (attr_done[dev] < lds) {
        if (hipFuncSetAttribute((const void*)mlp_chain_split_kernel<T, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        attr_done[dev] = 160 * 1024;
    }
    ProfScope ps(KID_CHAIN, flops, bytes, st);
    hipLaunchKernelGGL((mlp_chain_split_kernel<T, D>), dim3((unsigned)((p.M + 31) / 32)), dim3(256), lds, st, p);
    return lwdetr_check_launch();
}

template <typename T, int D, int KS0, bool RES, bool QP>
int launch_mlp_chain(const MlpChainParams& p, hipStream_t st, double flops, double bytes) {
    constexpr size_t lds = (size_t)32 * CH_PIECE_B + CH_VEC_MAX_B;
    static bool attr_done[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)mlp_chain_kernel<T, D, KS0, RES, QP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    ProfScope ps(KID_CHAIN, flops, bytes, st);
    hipLaunchKernelGGL((mlp_chain_kernel<T, D, KS0, RES, QP>), dim3((unsigned)((p.M + 127) / 128)), dim3(256), lds, st, p);
    return lwdetr_check_launch();
}

template <typename T, int D>
int dispatch_mlp_chain(const MlpChainParams& p, int k_in, bool res, bool qp, hipStream_t st, double flops, double bytes) {
    // row-per-wave form: D = 256 only (at D = 384 the rolled stage loop needs more than the 512 registers of a lone wave - hipcc 7.2
    // spills 280 - 950 bytes per lane - and the channel-split form serves every row count there)
    static_assert(D == 256, "row-per-wave form");
    if (k_in == 2 * D) return (res || qp) ? LWDETR_ERR_UNSUPPORTED : launch_mlp_chain<T, D, D / 8, false, false>(p, st, flops, bytes);
    if (k_in != D) return LWDETR_ERR_UNSUPPORTED;
    if (qp) return res ? launch_mlp_chain<T, D, D / 16, true, true>(p, st, flops, bytes) : LWDETR_ERR_UNSUPPORTED;
    return res ? launch_mlp_chain<T, D, D / 16, true, false>(p, st, flops, bytes) : launch_mlp_chain<T, D, D / 16, false, false>(p, st, flops, bytes);
}

// vec layout / piece count of a chain description (shared by the size helpers and the launch)
struct ChainLayout { long pieces, vec_floats; int bias_off[6], gam_off[6], bet_off[6], nt[6]; bool ok; };
ChainLayout chain_layout(const lwdetr_chain_desc* d) {
    ChainLayout L = {};
    if (!d || (d->D != 256 && d->D != 384) || d->nst < 1 || d->nst > 6 || (d->k_in != d->D && !(d->D == 256 && d->k_in == 2 * d->D))) return L;
    long off = 0, pieces = 0;
    for (int i = 0; i < d->nst; ++i) {
        const lwdetr_chain_stage& s = d->st[i];
        L.bias_off[i] = (int)off;
        if (s.kind == LWDETR_CHAIN_FULL) {
            const int k = (i == 0) ? d->k_in : d->D;
            L.nt[i] = d->D / 32;
            off += d->D;
            if (s.flags & LWDETR_CHAIN_LN) { L.gam_off[i] = (int)off; off += d->D; L.bet_off[i] = (int)off; off += d->D; }
            pieces += (long)(d->D / 32) * (k / 64);
        } else if (s.kind == LWDETR_CHAIN_SIDE) {
            if (s.n < 1 || (i == 0 && d->k_in != d->D)) return L;
            L.nt[i] = (s.n + 31) / 32;
            off += 32L * L.nt[i];
            pieces += (long)L.nt[i] * (d->D / 64);
        } else return L;
    }
    L.pieces = pieces + 2;
    L.vec_floats = (off * 4 + 4095) / 4096 * 4096 / 4;
    L.ok = L.vec_floats * 4 <= CH_VEC_MAX_B;
    return L;
}

}  // namespace

extern "C" long lwdetr_row_chain_pieces(const lwdetr_chain_desc* d) { const ChainLayout L = chain_layout(d); return L.ok ? L.pieces : -1; }
extern "C" long lwdetr_row_chain_vec_floats(const lwdetr_chain_desc* d) { const ChainLayout L = chain_layout(d); return L.ok ? L.vec_floats : -1; }

extern "C" int lwdetr_row_chain(const lwdetr_chain_desc* d, int dtype, void* hip_stream) {
    const ChainLayout L = chain_layout(d);
    if (!L.ok) return LWDETR_ERR_UNSUPPORTED;
    if (!d->in || !d->wstream || !d->vec || d->M < 0 || d->ld_in % 8 != 0 || ((uintptr_t)d->in | (uintptr_t)d->wstream | (uintptr_t)d->vec) % 16 != 0) return LWDETR_ERR_BAD_ARG;
    if (d->M == 0) return LWDETR_OK;
    if (d->k_in != d->D && d->st[0].kind != LWDETR_CHAIN_FULL) return LWDETR_ERR_BAD_ARG;
    if ((d->res && (d->ld_res % 8 != 0 || (uintptr_t)d->res % 16 != 0)) || (d->qpos && (d->ld_q % 8 != 0 || (uintptr_t)d->qpos % 16 != 0))) return LWDETR_ERR_BAD_ARG;
    MlpChainParams p = {};
    const auto bytes_of = [&](long ld) -> double { return (double)d->M * ld * 2.0; };
    if (bytes_of(d->ld_in) >= 2147483000.0 || (d->res && bytes_of(d->ld_res) >= 2147483000.0) || (d->qpos && bytes_of(d->ld_q) >= 2147483000.0)) return LWDETR_ERR_UNSUPPORTED;
    p.in = d->in; p.ld_in = d->ld_in; p.in_bytes = (unsigned)bytes_of(d->ld_in);
    p.res = d->res; p.ld_res = d->ld_res; p.res_bytes = d->res ? (unsigned)bytes_of(d->ld_res) : 0;
    p.qpos = d->qpos; p.ld_q = d->ld_q; p.q_bytes = d->qpos ? (unsigned)bytes_of(d->ld_q) : 0;
    p.wstream = d->wstream; p.vec = d->vec; p.vec_dpw = (int)(L.vec_floats * 4 / 4096); p.np = (int)L.pieces; p.M = d->M; p.nst = d->nst;
    double flops = 0.0, bytes = bytes_of(d->ld_in < d->k_in ? d->ld_in : d->k_in);
    bool uses_res = false, uses_q = false;
    for (int i = 0; i < d->nst; ++i) {
        const lwdetr_chain_stage& s = d->st[i];
        ChainStageK& k = p.st[i];
        k.kind = s.kind == LWDETR_CHAIN_FULL ? CS_FULL : CS_SIDE; k.nt = L.nt[i]; k.flags = s.flags; k.eps = s.eps;
        k.bias_off = L.bias_off[i]; k.gam_off = L.gam_off[i]; k.bet_off = L.bet_off[i];
        const bool stores = s.kind == LWDETR_CHAIN_SIDE || (s.flags & LWDETR_CHAIN_STORE);
        if (stores) {
            if (!s.out || s.ldo % 4 != 0 || (uintptr_t)s.out % 16 != 0 || bytes_of(s.ldo) >= 2147483000.0) return LWDETR_ERR_BAD_ARG;
            if (s.kind == LWDETR_CHAIN_FULL && (s.ldo % 8 != 0 || s.ldo < d->D)) return LWDETR_ERR_BAD_ARG;
            if (s.kind == LWDETR_CHAIN_SIDE && s.ldo < (s.n + 3) / 4 * 4) return LWDETR_ERR_BAD_ARG;
            k.out = s.out; k.ldo = s.ldo; k.out_bytes = (unsigned)bytes_of(s.ldo);
            bytes += (double)d->M * (s.kind == LWDETR_CHAIN_FULL ? d->D : s.n) * 2.0;
        }
        k.ncols = s.kind == LWDETR_CHAIN_SIDE ? s.n : d->D;
        if (s.kind == LWDETR_CHAIN_FULL) {
            if (s.flags & LWDETR_CHAIN_RES) uses_res = true;
            if (s.flags & LWDETR_CHAIN_ADDQ) uses_q = true;
            flops += 2.0 * d->M * d->D * (i == 0 ? d->k_in : d->D);
        } else {
            if (s.flags) return LWDETR_ERR_BAD_ARG;          // a SIDE stage takes no flags
            flops += 2.0 * d->M * d->D * 32.0 * L.nt[i];
        }
    }
    if ((uses_res && !d->res) || (uses_q && !d->qpos) || (uses_q && !uses_res && !d->res)) return LWDETR_ERR_BAD_ARG;
    if (uses_res) bytes += (double)d->M * d->D * 2.0;
    if (uses_q) bytes += (double)d->M * d->D * 2.0;
    hipStream_t st = (hipStream_t)hip_stream;
    const bool res = d->res != nullptr, qp = d->qpos != nullptr;
    // few rows: the channel-split form (32 rows per workgroup); many rows (or a 2 D-deep first stage): a wave per 32 rows
    const long split_rows = lwdetr_knob(KNOB_CHAIN_SPLIT_ROWS, 16384);      // <= 512 workgroups of 32 rows (two rounds of one per CU); measured: the heads chain of a
                                                                       // 16-image launch chain (14 400 rows) is 2-3 % of the step faster in this form (profiles/r4e_*)
    // D = 384: only the channel-split form (its stream is packed k-half-major: the two forms cannot read each other's streams)
    if (d->D == 384) {
        if (d->k_in != d->D) return LWDETR_ERR_UNSUPPORTED;
        return dtype == DT_F16 ? launch_mlp_chain_split<f16, 384>(p, st, flops, bytes) : dtype == DT_BF16 ? launch_mlp_chain_split<bf16, 384>(p, st, flops, bytes) : LWDETR_ERR_UNSUPPORTED;
    }
    if (d->k_in == d->D && d->M <= split_rows) {
        const int rc = dtype == DT_F16 ? launch_mlp_chain_split<f16, 256>(p, st, flops, bytes) : dtype == DT_BF16 ? launch_mlp_chain_split<bf16, 256>(p, st, flops, bytes) : LWDETR_ERR_UNSUPPORTED;
        if (rc != LWDETR_ERR_UNSUPPORTED) return rc;
    }
    if (dtype == DT_F16) return dispatch_mlp_chain<f16, 256>(p, d->k_in, res, qp, st, flops, bytes);
    if (dtype == DT_BF16) return dispatch_mlp_chain<bf16, 256>(p, d->k_in, res, qp, st, flops, bytes);
    return LWDETR_ERR_UNSUPPORTED;
}

extern "C" long lwdetr_enc_chain_vec_floats(int D, int k5) {
    const long f = (k5 ? 3L * D : 0) + 3L * D + 96 + 6L * D;
    return (f * 4 + 4095) / 4096 * 4096 / 4;
}
extern "C" long lwdetr_enc_chain_pieces(int D, int k5, int nl) {
    return (k5 ? (long)(D / 32) * (k5 / 64) : 0) + (long)(nl * (D / 32) + D / 32 + 3) * (D / 64) + 2;
}

extern "C" int lwdetr_enc_chain(const void* in, long ld_in, int k5, void* memory, void* om, void* cls, long ld_cls, float* cls_max,
                                void* const* values, int nl, const unsigned char* rowvalid, const unsigned char* notpad,
                                const void* wstream, const float* vec, long M, int D, int npix, int S, int lsi, long total_rows,
                                int ncls, float eps_p, float eps_e, int dtype, void* hip_stream) {
    if (!in || !om || !cls || !cls_max || !values || !rowvalid || !notpad || !wstream || !vec || M < 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    if (nl < 1 || nl > 6 || ncls < 1 || ncls > 96 || ld_cls < 96 || ld_cls % 8 != 0 || ld_in % 8 != 0 || npix <= 0 || S < npix || lsi < 0) return LWDETR_ERR_BAD_ARG;
    if (k5 && !memory) return LWDETR_ERR_BAD_ARG;
    if (((uintptr_t)in | (uintptr_t)om | (uintptr_t)cls | (uintptr_t)memory | (uintptr_t)wstream | (uintptr_t)vec) % 16 != 0) return LWDETR_ERR_BAD_ARG;
    if ((double)total_rows * (D > ld_cls ? D : ld_cls) * 2.0 >= 2147483000.0 || (double)M * ld_in * 2.0 >= 2147483000.0) return LWDETR_ERR_UNSUPPORTED;
    EncParams p = {};
    p.in = in; p.ld_in = ld_in; p.memory = k5 ? memory : nullptr; p.om = om; p.cls = cls; p.ld_cls = ld_cls; p.cls_max = cls_max;
    for (int i = 0; i < nl; ++i) {
        if (!values[i] || (uintptr_t)values[i] % 16 != 0) return LWDETR_ERR_BAD_ARG;
        p.values[i] = values[i];
    }
    p.nl = nl; p.rowvalid = rowvalid; p.notpad = notpad; p.wstream = wstream; p.vec = vec;
    p.np = (int)lwdetr_enc_chain_pieces(D, k5, nl);
    p.M = M; p.npix = npix; p.S = S; p.lsi = lsi; p.ncls = ncls; p.eps_p = eps_p; p.eps_e = eps_e;
    p.mem_bytes = (unsigned)((unsigned long)total_rows * D * 2ul);
    p.cls_bytes = (unsigned)((unsigned long)total_rows * ld_cls * 2ul);
    p.in_bytes = (unsigned)((unsigned long)M * ld_in * 2ul);
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F16: return dispatch_enc<f16>(p, D, k5, st);
        case DT_BF16: return dispatch_enc<bf16>(p, D, k5, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}
