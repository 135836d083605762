// Persistent large-tile GEMM for the compute-bound ViT GEMMs of the C = 768 model (round 6):  out = epilogue( A(M,K) x W(N,K)^T )
//
// Replaces, for the shapes it takes, gemm.hip's gemm_big_kernel (same 256 x 256 x 64 tile, same 8 waves of 128 x 64 on 32x32x16 MFMAs,
// same software-pipelined k-loop over (stage, 16-deep chunk) slots) - the reference call sites are the four Linear layers of a ViT block,
// models/backbone/vit.py:123-130 (qkv), :138 (proj), :217-218 + timm Mlp (fc1 -> GELU -> fc2), with their bias / GELU / LayerScale /
// residual / tap-copy / (B, heads, T, hd) layout epilogues fused. What is different, and why (profiles/r6a_*):
//
//  * A K = 768 tile of gemm_big_kernel spends ~10 of its ~29 us outside the k-loop with idle matrix pipes: the first two stages on their
//    way in (every workgroup of the chip in its prologue at once: ~11 B / clk / CU) and an epilogue that moves 256 KB of f32 through LDS
//    between workgroup barriers - through the SAME LDS the ring lives in, so nothing of the next tile can be in flight meanwhile.
//  * Here a workgroup is PERSISTENT: it walks its tiles (gridDim = CU count), and the DMA ring never stops at a tile boundary - stage
//    nk of a tile IS stage 0 of the next one. The k-loop's own issue slots (A pieces of stage kt + 2 in the last slot of stage kt, W pieces
//    of stage kt + 1 in its first slot) therefore fetch the next tile's first stages during the current tile's last two steps, and the
//    fragments of its first chunk are in registers when the epilogue starts.
//  * The epilogue goes STRAIGHT FROM THE ACCUMULATORS to memory: a v_permlane32_swap between the two half-waves turns a lane's 4 + 4
//    outputs (two 8-row groups of the 32x32 accumulator layout) into 8 consecutive outputs = one 16-byte store per lane, 32 rows x
//    32 bytes per instruction (T21 of the CDNA guide; vitblock.hip's vb_rows8). No LDS, no barrier: every wave finishes its own
//    128 x 64 block on its own and falls back into the k-loop; the ring's next stages land under it.
//  * Only scalar state (tile index, ring parity) is live across the epilogue besides the 24 fragment registers of the next chunk; the
//    round-2 persistent variant kept its tile-loop state in vector registers around an LDS epilogue and spilled ~60 of them.
//  * DMA addresses: lane part (row-in-piece x lda + swizzled k-slot, 32-bit, fixed for the whole launch) in ONE vector register per
//    operand, everything that changes with tile / stage / piece is scalar (global_load_lds with an SGPR base).
//
// Arithmetic: gemm_big_kernel's MFMA sequence over k with the accumulators starting at the bias instead of zero, and ONE multiplication by
// scale * gamma[n] where that kernel multiplies twice - equal to it within an f32 rounding or two (tests/test_gpu_kernels.py bounds the difference).
// Own translation unit + own epilogue copy on purpose (round-5 rule, profiles/r5g_*): nothing here can move gemm.hip's register allocation.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

template <typename T> struct PtMma;
template <> struct PtMma<f16> {
    static __device__ __forceinline__ f32x16 k16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct PtMma<bf16> {
    static __device__ __forceinline__ f32x16 k16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

constexpr int PT_BM = 256, PT_BN = 256, PT_KB = 64, PT_NW = 8;
constexpr int PT_EPC = 8, PT_RP = 8, PT_KC = PT_KB / 16;            // elements per 16 bytes, rows per DMA piece, 16-deep chunks per stage
constexpr int PT_WM = 128, PT_WN = 64, PT_TM = 4, PT_TN = 2;        // wave tile 128 x 64 = 4 x 2 MFMA tiles; wave grid 2 (rows) x 4 (columns)
constexpr int PT_A_MY = PT_BM / PT_RP / PT_NW, PT_B_MY = PT_BN / PT_RP / PT_NW;      // DMA pieces per wave, stage and operand: 4 + 4
constexpr int PT_STAGE = (PT_BM + PT_BN) * PT_KB;                  // elements per ring stage (64 KB); ring = 2 stages = 128 KB

template <typename T, int ACT> __device__ __forceinline__ float pt_act(float x) {
    if (ACT == ACT_GELU) return gelu_for<T>(x);
    if (ACT == ACT_SILU) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
    if (ACT == ACT_RELU) return x > 0.f ? x : 0.f;
    return x;
}

constexpr int PT_NMAX = 4096;                                       // columns whose bias / scale vectors fit the LDS behind the ring (2 x 16 KB)

// Phase timing for kernel tuning (a private build with -DLWDETR_PT_TIMING=<workgroup>, tools/pt_timing.py; never in the product): per wave of
// that workgroup, 10 ns ticks summed over its tiles - k-loop, epilogue, the first stage wait after an epilogue - and its tile count.
#ifdef LWDETR_PT_TIMING
__device__ unsigned long long g_pt_timing[8][8];
extern "C" int lwdetr_debug_pt_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pt_timing), sizeof(g_pt_timing)) == hipSuccess ? 0 : 1;
}
#define PT_NOW() __builtin_amdgcn_s_memrealtime()
#endif

template <typename T>
__global__ __launch_bounds__(512) void gemm_pt_kernel(const lwdetr_gemm_desc d, const int tiles_n, const int ntiles, const int skew_ticks) {
    typedef typename Vec<T>::v8 V8;
    constexpr int BM = PT_BM, BN = PT_BN, KB = PT_KB, NW = PT_NW, EPC = PT_EPC, RP = PT_RP, KC = PT_KC;
    constexpr int WM = PT_WM, WN = PT_WN, TM = PT_TM, TN = PT_TN, A_MY = PT_A_MY, B_MY = PT_B_MY, STAGE = PT_STAGE;
    constexpr int PIECES = A_MY + B_MY;
    extern __shared__ __attribute__((aligned(16))) char pt_smem[];
    const T* smem = (const T*)pt_smem;
    float* lds_bias = (float*)(pt_smem + 2 * STAGE * sizeof(T));          // [PT_NMAX] bias of column n (0 without one), then [PT_NMAX] scale * gamma
    float* lds_cs = lds_bias + PT_NMAX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    // ---- this workgroup's tile sequence. Workgroup b runs on XCD b % 8 (observed, used for speed only): XCD x owns a contiguous eighth of the
    // row-panel-major tile order, its workgroups walk that range side by side - the tiles in flight on an XCD at any time are neighbours
    // (one A row panel is fetched into that XCD's L2 once for all its column tiles).
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, wslot = bid >> 3;
    const int wgx = (nwg - xcd + 7) >> 3;                                   // workgroups on this XCD
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int t_begin = xcd * q8 + (xcd < r8 ? xcd : r8), t_len = q8 + (xcd < r8 ? 1 : 0);
    int tl = wslot;                                                         // position inside the XCD's range
    if (tl >= t_len) return;

#ifdef LWDETR_PT_TIMING
    const int pt_abl = skew_ticks >> 24;          // timing builds: ablations (results are wrong): 1 = no epilogue stores, 2 = no residual loads
#define PT_ABL(b) (pt_abl & (b))
#else
#define PT_ABL(b) 0
#endif
    // Start skew (tuning; 0 by default): workgroup slot s of its XCD starts s / 32 of the window late (10 ns ticks of the constant 100 MHz counter)
    if ((skew_ticks & 0xffffff) > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)((unsigned)wslot * (unsigned)(skew_ticks & 0xffffff)) / (unsigned)(wgx > 1 ? wgx : 1);
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    // ---- per-column epilogue vectors into LDS, once per workgroup: bias[n] and scale * gamma[n] of the segment that owns column n. The
    // accumulators of a tile START at the bias (the C operand of its first MFMAs), the scale vector is read per 8-column run in the epilogue.
    for (int n = tid; n < d.N; n += NW * 64) {
        int s_ = 0;
#pragma unroll
        for (int s = 1; s < 3; ++s) if (s < d.nseg && n >= d.seg[s].n_begin) s_ = s;
        const float* bp = s_ == 0 ? d.seg[0].bias : (s_ == 1 ? d.seg[1].bias : d.seg[2].bias);
        const float* gp = s_ == 0 ? d.seg[0].gamma : (s_ == 1 ? d.seg[1].gamma : d.seg[2].gamma);
        const float sc = s_ == 0 ? d.seg[0].scale : (s_ == 1 ? d.seg[1].scale : d.seg[2].scale);
        const int nb_ = s_ == 0 ? 0 : (s_ == 1 ? d.seg[1].n_begin : d.seg[2].n_begin);
        lds_bias[n] = bp ? bp[n - nb_] : 0.f;
        lds_cs[n] = gp ? sc * gp[n - nb_] : sc;
    }
    __syncthreads();

    const int nk = d.K / KB;
    const char* __restrict__ Ab = (const char*)d.A;
    const char* __restrict__ Wb = (const char*)d.W;

    // ---- DMA: piece kk of an operand covers tile rows 64 kk + 8 wave + lane / 8 (the 16-byte k-slot lane % 8, un-swizzled with the row's key,
    // is what the lane fetches); lane-linear in LDS. The lane part of the address is one 32-bit byte offset per operand for the whole launch;
    // everything that changes with tile / stage / piece is scalar: base(tile) + 128 kt + kk * (64 rows).
    const int prow = lane >> 3, pslot = lane & 7;
    const int row0 = RP * (tid >> 6) + prow;
    const int swz = (pslot ^ ((row0 >> 1) & 7)) * EPC;
    const unsigned voff_a = (unsigned)(((long)row0 * d.lda + swz) * (long)sizeof(T));
    const unsigned voff_w = (unsigned)(((long)row0 * d.K + swz) * (long)sizeof(T));
    const unsigned lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)pt_smem + (unsigned)wave * 1024u;
    const long stride_a = 64L * d.lda * (long)sizeof(T), stride_w = 64L * d.K * (long)sizeof(T);
    // piece k (0 .. 3: A, 4 .. 7: W) of the stage whose operand bases are pa / pw, into ring buffer sb
    auto issue = [&](const char* pa, const char* pw, int sb, int k) {
        const bool is_a = k < A_MY;
        const int kk = is_a ? k : k - A_MY;
        const char* base = (is_a ? pa + kk * stride_a : pw + kk * stride_w);
        const unsigned dst = lds_wave + (unsigned)(sb * STAGE * (int)sizeof(T)) + (unsigned)((is_a ? 0 : BM * KB * (int)sizeof(T)) + kk * NW * 1024);
        if (is_a) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff_a), "s"(base) : "memory");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff_w), "s"(base) : "memory");
    };
    // The last row tile of a ragged M loads (and multiplies) the LAST 256 rows of A instead of rows past M: every address stays affine in
    // (tile, piece, stage), and its epilogue stores only the rows from m0 on (the others belong to the previous tile). Operand bases are
    // recomputed from the 32-bit tile coordinates once per k-step (a dozen scalar instructions) rather than kept as 64-bit pointers per tile:
    // the scalar file is as full as the vector file here.
    auto a_first = [&](int tm0) { return tm0 + BM <= d.M ? tm0 : d.M - BM; };
    auto a_base = [&](int ma_, int kt_) { return Ab + ((long)ma_ * d.lda + (long)kt_ * KB) * (long)sizeof(T); };
    auto w_base = [&](int tn0, int kt_) { return Wb + ((long)tn0 * d.K + (long)kt_ * KB) * (long)sizeof(T); };

    // fragment reads: lane -> row (lane & 31) of a 32-row MFMA tile, 16-byte k-slot 2 c + (lane >> 5) of chunk c, swizzled with the row's key
    int pofs[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) pofs[c] = ((2 * c + h) ^ ((m >> 1) & 7)) * EPC;
    const int arow = (wm * WM + m) * KB, brow = BM * KB + (wn * WN + m) * KB;

    // ---- first tile: stages 0 and 1 requested, stage 0 awaited
    int tile = t_begin + tl;
    int tmi = tile / tiles_n;
    int m0 = tmi * BM;
    int n0 = (tile - tmi * tiles_n) * BN;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < PIECES; ++k) issue(a_base(a_first(m0), s), w_base(n0, s), s, k);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
    __builtin_amdgcn_s_barrier();
    int gs = 0;                          // ring slot of the current tile's stage 0
    bool stores_behind = false;          // an epilogue that issued its full complement of stores (>= 16 per wave) lies behind the ring's newest pieces
#ifdef LWDETR_PT_TIMING
    unsigned long long tt_loop = 0, tt_epi = 0, tt_wait0 = 0, tt_tiles = 0;
    const unsigned long long tt_begin = PT_NOW();
#endif

    for (;;) {
        // (loop-carried scalars restated as wave-uniform: a phi behind the epilogue's lane-masked stores is "divergent" to hipcc's analysis, and the
        // DMA statements need their bases in scalar registers)
        tl = __builtin_amdgcn_readfirstlane(tl); m0 = __builtin_amdgcn_readfirstlane(m0); n0 = __builtin_amdgcn_readfirstlane(n0);
        gs = __builtin_amdgcn_readfirstlane(gs);
        // the next tile of this workgroup (scalar state only)
        const int tl_n = tl + wgx;
        const bool has_next = tl_n < t_len;
        const int tile_n = t_begin + (has_next ? tl_n : tl), tmi_n = tile_n / tiles_n;       // no next tile: the current one again (dummy re-reads)
        const int m0n = tmi_n * BM;
        const int n0n = (tile_n - tmi_n * tiles_n) * BN;
        const int ma = a_first(m0), man = a_first(m0n);                   // first rows this tile / the next one actually multiply

        int si = 0;
#pragma unroll
        for (int s = 1; s < 3; ++s) if (s < d.nseg && n0 >= d.seg[s].n_begin) si = s;
        const lwdetr_gemm_seg& sg = d.seg[si];

        // The k-loop is instantiated once per MFMA operand order (ROW: lanes own a row, registers run along the columns - LINEAR / HEADS;
        // COL: lanes own a column, registers run along the rows - HEADS_T, the transposed V of attention).
        auto body = [&](auto col_tag) {
            constexpr bool COL = decltype(col_tag)::value;
#ifdef LWDETR_PT_TIMING
            const unsigned long long t_a = PT_NOW();
#endif
            // The accumulators start at the bias of their column: the first MFMA of every accumulator tile takes bvec[j] as its C operand.
            // 32x32 accumulator: register 4 q + r of lane (c = lane & 31, hi) is element (8 q + 4 hi + r, c) of D; ROW: D = W tile x rows^T
            // (element index = column), COL: D = rows x W tile^T (lane = column).
            f32x16 bvec[TN];
            {
                const float* bl = lds_bias + n0 + wn * WN;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (COL) {
                        const float b = bl[j * 32 + m];
#pragma unroll
                        for (int e = 0; e < 16; ++e) bvec[j][e] = b;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 b = *(const f32x4*)(bl + j * 32 + q * 8 + h * 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) bvec[j][4 * q + r] = b[r];
                        }
                    }
                }
            }
            f32x16 acc[TN][TM];
            // the fragments of chunk 0 of stage 0 (landed and barrier-ed: by the prologue, or in the last slot of the previous tile - which read
            // them once already; they are read again here rather than kept through the epilogue, where their 24 registers are needed)
            V8 xf[2][TM], wf[2][TN];
            {
                const T* A0 = smem + (gs & 1) * STAGE;
#pragma unroll
                for (int i = 0; i < TM; ++i) xf[0][i] = *(const V8*)(A0 + arow + i * 32 * KB + pofs[0]);
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[0][j] = *(const V8*)(A0 + brow + j * 32 * KB + pofs[0]);
            }
            // Software pipeline over (stage, chunk) slots as in gemm_big_kernel: slot (kt, c) multiplies chunk c out of one fragment buffer
            // while it reads the next chunk - (kt, c + 1), or chunk 0 of stage kt + 1 in the last slot - into the other, one ds_read after
            // each MFMA; the stage barrier sits in front of the LAST slot of a stage (all its fragments are in registers: its buffer is free).
            // Stage numbers run on past nk into the next tile: the last two steps of a tile fetch the first two stages of the next one, ahead
            // of the epilogue.
            // DMA schedule (gemm_big_kernel's; all 8 pieces in the last slot measured 3 % slower, profiles/r6a_*): the W pieces of stage kt + 1 go out in
            // slot 0 of stage kt, the A pieces of stage kt + 2 in its last slot; step 0 of a tile is peeled - its W pieces (stage 1) went out before
            // the tile began (first tile: with the prologue; later tiles: right behind the previous tile's k-loop, ahead of its epilogue).
            // at tile start: A pieces up to stage 1 and W pieces up to stage 1 are out -> step 0 issues A of stage 2, step 1 W of stage 2
            const char* pa_next = a_base(man, 0);
            const char* pw_next = w_base(n0n, 0);
            const char* pa_iss = nk > 2 ? a_base(ma, 2) : pa_next;
            const char* pw_iss = w_base(n0, 1);          // (step 0 issues no W pieces: it only moves this pointer on to stage 2)
            auto step = [&](int kt, auto first_tag) {
                constexpr bool FIRST = decltype(first_tag)::value;
                const int sb = (gs + kt) & 1;
                const T* As = smem + sb * STAGE;
                const T* An = smem + (sb ^ 1) * STAGE;
                // operand bases of the pieces this step issues (A: stage kt + 2, W: stage kt + 1; past nk: the next tile's): running 64-bit
                // scalars, 128 bytes on per step, switched to the next tile's base at its stage 0 (recomputing them from the tile coordinates
                // cost ~50 scalar instructions at every stage barrier)
                const char* pa = pa_iss;
                const char* pw = pw_iss;
                pa_iss = kt + 3 == nk ? pa_next : pa_iss + KB * (int)sizeof(T);
                pw_iss = kt + 2 == nk ? pw_next : pw_iss + KB * (int)sizeof(T);
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    constexpr int NM = TN * TM;
                    const bool last = c == KC - 1;
                    if (last) {
#ifdef LWDETR_PT_TIMING
                        const unsigned long long t_w = PT_NOW();
#endif
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the last fragments of stage kt are in registers
                        // stage kt + 1 has landed (this wave's pieces). Behind a full epilogue its >= 16 stores are the youngest operations of the
                        // queue and need not be waited for (vmcnt retires in order: at most 16 outstanding = every piece has landed)
                        if (FIRST && stores_behind) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
#ifdef LWDETR_PT_TIMING
                        if (FIRST) tt_wait0 += PT_NOW() - t_w;
#endif
                    }
                    const T* Ar = last ? An : As;
                    const int po = pofs[(c + 1) % KC];
                    const int nb = (c + 1) & 1, cb = c & 1;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int idx = 0; idx < NM; ++idx) {
                        const int j = idx / TM, i = idx % TM;
                        const f32x16 cin = FIRST && c == 0 ? bvec[j] : acc[j][i];
                        acc[j][i] = COL ? PtMma<T>::k16(xf[cb][i], wf[cb][j], cin) : PtMma<T>::k16(wf[cb][j], xf[cb][i], cin);
                        // one fragment read after each MFMA, in the order the next slot's MFMAs want them: x0, w0, x1 .. x3, w1
                        if (idx == 0) xf[nb][0] = *(const V8*)(Ar + arow + po);
                        else if (idx == 1) wf[nb][0] = *(const V8*)(Ar + brow + po);
                        else if (idx < TM + 1) xf[nb][idx - 1] = *(const V8*)(Ar + arow + (idx - 1) * 32 * KB + po);
                        else if (idx < TM + TN) wf[nb][idx - TM] = *(const V8*)(Ar + brow + (idx - TM) * 32 * KB + po);
                        // DMA issue, branch-free (a branch would cut the pinned MFMA / ds_read stream into basic blocks). Without a next tile the
                        // last two steps re-read this tile's first stages into the freed buffers (nobody reads them, the exit drains them).
                        if (last && idx < A_MY) issue(pa, pw, sb, idx);
                        if (c == 0 && !FIRST && idx < B_MY) issue(pa, pw, sb ^ 1, A_MY + idx);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    static_assert(PIECES <= NM, "one DMA piece per MFMA at most");
                }
            };
            step(0, std::true_type{});
            for (int kt = 1; kt < nk; ++kt) step(kt, std::false_type{});
            {                      // the W pieces of the next tile's stage 1 (their buffer - stage nk - 1's - was freed by the last barrier), ahead of the epilogue
#pragma unroll
                for (int k = 0; k < B_MY; ++k) issue(pw_iss, pw_iss, (gs + nk + 1) & 1, A_MY + k);
            }
#ifdef LWDETR_PT_TIMING
            const unsigned long long t_b = PT_NOW();
#endif

            // ---- epilogue, register-direct: per 8-output run 4 half-wave exchanges, [activation], [x column scale], [+ residual], 4 packs, one
            // 16-byte store. An exchange (v_permlane32_swap, in place) between accumulator registers 8 p + r and 8 p + 4 + r leaves lanes 0-31
            // with [own q = 2p | partner's q = 2p] and lanes 32-63 with [partner's q = 2p + 1 | own q = 2p + 1] in registers 8 p .. 8 p + 7:
            // outputs 16 p + 8 h + 0..7 along the register axis of the 32 x 32 tile (T21 of the CDNA guide).
            // The lane-dependent addressing is derived from an OPAQUE copy of the lane id made here, so that hipcc cannot hoist it (row offsets:
            // ~20 registers) out of the tile loop and across the k-loop, where every register is taken.
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int m = lane_e & 31, h = lane_e >> 5;
            const int nlw = n0 - sg.n_begin + wn * WN;                  // first column of this wave's block inside the segment
            const float* csl = lds_cs + n0 + wn * WN;
            const bool has_cs = sg.gamma != nullptr || sg.scale != 1.f; // wave-uniform
            // the lane's 8 consecutive outputs of accumulator registers 8 p .. 8 p + 7 (scalar temporaries on purpose: element-wise writes into a
            // 16-register accumulator tuple make hipcc spill hundreds of registers). s_nop 1: an operand may have been written by a VALU copy just in
            // front of the statement - 2 wait states, which hipcc does not insert for inline assembly.
            auto run8 = [&](const f32x16& a, int p, float (&x)[8]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float lo = a[8 * p + r], hi = a[8 * p + 4 + r];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
                    x[r] = lo; x[4 + r] = hi;
                }
            };
            auto finish = [&](auto act_tag) {
                constexpr int ACT = decltype(act_tag)::value;
                if constexpr (!COL) {
                    T* __restrict__ out = (T*)sg.out;
                    T* __restrict__ out2 = (T*)sg.out2;
                    const T* __restrict__ res = (const T*)sg.res;
                    const bool heads = sg.mode == LWDETR_OUT_HEADS;
                    const int hd_sh = 31 - __builtin_clz((unsigned)(sg.p1 > 0 ? sg.p1 : 1));
                    const unsigned ldo = (unsigned)sg.ldo, ldres = (unsigned)sg.ldres, ld2 = (unsigned)sg.ld2;
                    const int Tp = sg.p0, hdim = sg.p1, nh = sg.p2;
                    // rows of the lane: ma + wm WM + 32 i + (lane & 31). HEADS: a 256-row tile crosses at most one image boundary (Tp >= 256).
                    // Element offsets fit 32 bits (lwdetr_gemm_pt_try).
                    unsigned roff[TM];
                    int mrow[TM];
                    const int b0 = heads ? ma / Tp : 0;
                    const int t0 = heads ? ma - b0 * Tp : 0;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int lr = wm * WM + i * 32 + m;
                        mrow[i] = ma + lr;
                        if (heads) {
                            int t = t0 + lr, b = b0;
                            if (t >= Tp) { t -= Tp; ++b; }
                            roff[i] = (unsigned)((b * nh * Tp + t) * hdim);
                        } else roff[i] = (unsigned)mrow[i] * ldo;
                    }
                    // Column groups g = (j, p): the lane's 8 columns nl .. nl + 7 (inside one head: hd % 8 == 0). ALL 16 residual runs of the wave are
                    // requested before its first store, unconditionally (every row this tile multiplies exists): vmcnt retires in order, so a wait
                    // for a load that was issued BEHIND a store waits for that store's acknowledgement as well - with the loads of group g + 1
                    // issued between the stores of groups g - 1 and g the epilogue of a residual tile took 15 us instead of 4 (a store round trip
                    // per run, profiles/r6a_*); and a load under a per-row test makes hipcc branch around it and drain the queue per element.
                    // FULL (wave-uniform): every row is stored, straight-line code.
                    auto rows = [&](auto full_tag) {
                        constexpr bool FULL = decltype(full_tag)::value;
                        auto nl_of = [&](int g) { return nlw + (g >> 1) * 32 + (g & 1) * 16 + h * 8; };
                        V8 rv[2 * TN][TM];
                        if (res && !PT_ABL(2)) {
#pragma unroll
                            for (int g = 0; g < 2 * TN; ++g)
#pragma unroll
                                for (int i = 0; i < TM; ++i) rv[g][i] = *(const V8*)(res + (unsigned)mrow[i] * ldres + (unsigned)nl_of(g));
                        }
#pragma unroll
                        for (int g = 0; g < 2 * TN; ++g) {
                            const int j = g >> 1, p = g & 1;
                            const int nl = nl_of(g);
                            f32x4 c0v = {1.f, 1.f, 1.f, 1.f}, c1v = c0v;
                            if (has_cs) { c0v = *(const f32x4*)(csl + j * 32 + p * 16 + h * 8); c1v = *(const f32x4*)(csl + j * 32 + p * 16 + h * 8 + 4); }
                            unsigned coff = (unsigned)nl;
                            if (heads) { const int hh = nl >> hd_sh; coff = (unsigned)(hh * Tp * hdim + (nl - (hh << hd_sh))); }
#pragma unroll
                            for (int i = 0; i < TM; ++i) {
                                float x[8];
                                run8(acc[j][i], p, x);
                                if (ACT != ACT_NONE) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) x[e] = pt_act<T, ACT>(x[e]);
                                }
                                if (has_cs) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { x[e] *= c0v[e]; x[4 + e] *= c1v[e]; }
                                }
                                if (res) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) x[e] += to_f32<T>(rv[g][i][e]);
                                }
                                V8 o;
#pragma unroll
                                for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(x[e]);
                                if (PT_ABL(1)) asm volatile("" :: "v"(o));
                                else if (FULL || mrow[i] >= m0) {
                                    *(V8*)(out + roff[i] + coff) = o;
                                    if (out2) *(V8*)(out2 + (unsigned)mrow[i] * ld2 + (unsigned)nl) = o;
                                }
                            }
                        }
                    };
                    if (ma == m0) rows(std::true_type{});
                    else rows(std::false_type{});
                } else {
                    // HEADS_T: out[((b heads + hh) hd + dd) Tp + t]; the lane owns channel nlw + 32 j + (lane & 31), its 8 consecutive tokens are
                    // rows ma + wm WM + 32 i + 16 p + 8 h .. + 7 (Tp % 8 == 0 and M % 8 == 0: a run never straddles an image or the tile's first row)
                    T* __restrict__ out = (T*)sg.out;
                    const int hd_sh = 31 - __builtin_clz((unsigned)sg.p1);
                    const int Tp = sg.p0, hdim = sg.p1, nh = sg.p2;
                    const float scale = sg.scale;
                    const int b0 = ma / Tp, t0 = ma - b0 * Tp;
                    auto cols = [&](auto full_tag) {
                    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int nl = nlw + j * 32 + m;
                        const int hh = nl >> hd_sh, dd = nl - (hh << hd_sh);
                        const unsigned ocol = (unsigned)((hh * hdim + dd) * Tp);
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int p = 0; p < 2; ++p) {
                                const int lr = wm * WM + i * 32 + p * 16 + h * 8;
                                int t = t0 + lr, b = b0;
                                if (t >= Tp) { t -= Tp; ++b; }
                                float x[8];
                                run8(acc[j][i], p, x);
                                V8 o;
#pragma unroll
                                for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(pt_act<T, ACT>(x[e]) * scale);
                                if (PT_ABL(1)) asm volatile("" :: "v"(o));
                                else if (FULL || ma + lr >= m0) *(V8*)(out + ocol + (unsigned)(b * nh * hdim * Tp + t)) = o;
                            }
                    }
                    };
                    if (ma == m0) cols(std::true_type{});
                    else cols(std::false_type{});
                }
            };
            const int act = sg.act;
            if (act == ACT_NONE) finish(std::integral_constant<int, ACT_NONE>{});
            else if (act == ACT_GELU) finish(std::integral_constant<int, ACT_GELU>{});
            else if (act == ACT_SILU) finish(std::integral_constant<int, ACT_SILU>{});
            else finish(std::integral_constant<int, ACT_RELU>{});
#ifdef LWDETR_PT_TIMING
            tt_loop += t_b - t_a; tt_epi += PT_NOW() - t_b; ++tt_tiles;
#endif
        };
        if (sg.mode == LWDETR_OUT_HEADS_T) body(std::true_type{});
        else body(std::false_type{});

        if (!has_next) break;            // (the dummy pieces of the last two steps are drained below)
        stores_behind = ma == m0;        // every row of the tile was stored: 16 (32 with a tap copy) stores per wave, all younger than the ring's pieces
        gs = (gs + nk) & 1;
        tl = tl_n; m0 = m0n; n0 = n0n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may still be on its way into this workgroup's LDS when it is handed on
#ifdef LWDETR_PT_TIMING
    if (blockIdx.x == LWDETR_PT_TIMING && lane == 0) {
        unsigned long long* o = g_pt_timing[wave];
        o[0] = tt_loop; o[1] = tt_epi; o[2] = tt_wait0; o[3] = tt_tiles; o[4] = PT_NOW() - tt_begin;
    }
#endif
}

constexpr int PT_LDS_BYTES = 2 * PT_STAGE * 2 + 2 * PT_NMAX * 4;         // ring + bias / scale vectors = 160 KB
int g_pt_mode = -1;
int g_pt_skew = -1;
long g_pt_launches = 0;        // launches of the persistent kernel by this process (tests assert which kernel served a shape)

template <typename T>
int pt_launch(const lwdetr_gemm_desc& d, hipStream_t st, bool& taken) {
    static signed char state[16] = {};          // per device: 0 = not asked yet, 1 = granted, -1 = refused
    static int ncu[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_OK;
    if (state[dev] == 0) {
        state[dev] = hipFuncSetAttribute((const void*)gemm_pt_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS_BYTES) == hipSuccess ? 1 : -1;
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        ncu[dev] = n;
    }
    if (state[dev] < 0) { (void)hipGetLastError(); return LWDETR_OK; }
    const int tiles_n = d.N / PT_BN;
    const long ntiles = ((d.M + PT_BM - 1) / PT_BM) * tiles_n;
    const int grid = (int)(ntiles < ncu[dev] ? ntiles : ncu[dev]);
    int skew = g_pt_skew >= 0 ? g_pt_skew : (int)lwdetr_knob(KNOB_GEMM_PT_SKEW, 0);      // LWDETR_GEMM_PT_SKEW: start-skew window in 10 ns ticks (tuning)
    if (skew < 0) skew = 0;
    hipLaunchKernelGGL((gemm_pt_kernel<T>), dim3((unsigned)grid), dim3(512), PT_LDS_BYTES, st, d, tiles_n, (int)ntiles, skew);
    taken = true;
    ++g_pt_launches;
    return lwdetr_check_launch();
}

}  // namespace

extern "C" void lwdetr_gemm_pt_tuning(int pt_mode) { g_pt_mode = pt_mode < 0 ? -1 : (pt_mode & 0xff); g_pt_skew = pt_mode < 0x100 ? -1 : (pt_mode >> 8); }
extern "C" long lwdetr_gemm_pt_count(void) { return g_pt_launches; }

// Shapes the persistent kernel takes (everything else stays on gemm.hip's kernels): 16-bit, plain A (no A2, no LayerNorm fold), K % 64 == 0, whole
// 256-column tiles inside every segment, M % 8 == 0, LINEAR / HEADS / HEADS_T segments without row masks or a periodic residual, head dimension a
// power of two >= 8, >= 256 tokens per image - and, by default, the sizes at which the 256-row tile pays at all (gemm.hip: try_launch_big).
int lwdetr_gemm_pt_try(const lwdetr_gemm_desc& d, int dtype, hipStream_t st, bool& taken) {
    taken = false;
    const int mode = g_pt_mode >= 0 ? g_pt_mode : (int)lwdetr_knob(KNOB_GEMM_PT, 1);    // LWDETR_GEMM_PT: 0 = never, 1 = default shapes, 2 = whenever legal
    if (!mode || (dtype != DT_F16 && dtype != DT_BF16)) return LWDETR_OK;
    if (d.a_mode != LWDETR_A_PLAIN || d.A2 || d.K % PT_KB != 0 || d.K < 2 * PT_KB || d.N % PT_BN != 0 || d.N > PT_NMAX || d.M % 8 != 0 || d.M < PT_BM || d.lda % 8 != 0 ||
        ((size_t)d.A & 15) != 0 || ((size_t)d.W & 15) != 0 || (long)64 * d.lda * 2 + 256 >= (1L << 31)) return LWDETR_OK;
    for (int s = 0; s < d.nseg; ++s) {
        const lwdetr_gemm_seg& g = d.seg[s];
        const int n_end = g.n_end < d.N ? g.n_end : d.N;
        if (g.n_begin % PT_BN != 0 || n_end % PT_BN != 0 || g.rowmask || g.ln_stats || g.res_mod > 0 || ((size_t)g.out & 15) != 0) return LWDETR_OK;
        if (g.mode == LWDETR_OUT_LINEAR) {
            if (g.ldo % 8 != 0 || (g.res && (g.ldres % 8 != 0 || ((size_t)g.res & 15) != 0)) || (g.out2 && (g.ld2 % 8 != 0 || ((size_t)g.out2 & 15) != 0))) return LWDETR_OK;
        } else if (g.mode == LWDETR_OUT_HEADS || g.mode == LWDETR_OUT_HEADS_T) {
            if (g.p1 < 8 || (g.p1 & (g.p1 - 1)) != 0 || g.p0 < PT_BM || g.p0 % 8 != 0 || g.res || g.out2 || g.gamma) return LWDETR_OK;
        } else return LWDETR_OK;
    }
    for (int s = 0; s < d.nseg; ++s) {          // 32-bit element offsets in the epilogue
        const lwdetr_gemm_seg& g = d.seg[s];
        const long lim = 1L << 31;
        if (g.mode == LWDETR_OUT_LINEAR ? ((long)d.M * g.ldo >= lim || (g.res && (long)d.M * g.ldres >= lim) || (g.out2 && (long)d.M * g.ld2 >= lim))
                                        : (long)d.M * g.p1 * g.p2 >= lim) return LWDETR_OK;
    }
    const long tiles = ((d.M + PT_BM - 1) / PT_BM) * (d.N / PT_BN);
    if (tiles < 8) return LWDETR_OK;
    if (mode == 1 && !(d.K >= 384 && (d.M >= 16384 || (d.K >= 960 && tiles >= 96)))) return LWDETR_OK;
    // Launches with a residual stay on gemm_big_kernel by default: the 16 residual runs of a wave (all requested ahead of its first store) cost the
    // register-direct epilogue 9-12 us per tile against 4-6 without one - attention projection 107 vs 102 us, fc2 288-296 vs 276-286 us at xlarge
    // (profiles/r6a_*: touching the rows into L2 two steps ahead changed nothing; neither did a start skew). QKV (-20 %) and fc1 (-2 %) take it.
    if (mode == 1)
        for (int s = 0; s < d.nseg; ++s) if (d.seg[s].res) return LWDETR_OK;
    return dtype == DT_F16 ? pt_launch<f16>(d, st, taken) : pt_launch<bf16>(d, st, taken);
}
