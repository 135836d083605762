// Fused transformer MLP for gfx950:  x <- x + gamma2 * ( fc2( GELU( fc1( LN(x) ) ) ) )         (one launch)
//
// Replaces, per ViT block, LayerNorm + Linear(C,4C) + GELU + Linear(4C,C) + LayerScale + residual
// (reference models/backbone/vit.py:217-218 with timm Mlp): the 4C-wide hidden activation never leaves the CU.
// HBM traffic per block drops from  read x, write LN(x), read LN(x), write h(4C), read h(4C), read x, write x
// to  read x, write x  (+ weights through L2).
//
// Structure (512 threads = 8 waves = ONE workgroup per CU; each wave owns one tile of 16*TT tokens, all eight share the
// weight tiles in LDS - the weights are re-streamed once per workgroup, and a CU can only pull ~10-12 B/clk through
// its load path, so tokens-per-workgroup is what sets the streaming cost. Tiles are dealt evenly over <= #CU workgroups):
//   prologue : wave loads its token rows as MFMA B-operand fragments (lane: token l15, 8 channels per k-chunk) and
//              normalises them in registers (two-pass f32 statistics; LN's affine is folded into W1/b1 on the host).
//   hidden loop over chunks of 32 hidden units, double-buffered LDS tiles W1c[32][C] and W2c[C][32] (chunk-major
//              repacked on the host so every tile is one contiguous 12 KB block), one barrier per chunk:
//     step 1 : D1[hidden][token] = W1c * LN(x)^T      (A = W1c rows, B = x fragments)         2 x C/32 MFMAs / tile
//     GELU   : in registers on the accumulator layout (lane: 4 hidden x 1 token)
//     step 2 : D2[n][token] += W2c[n][hidden] * H      H is ALREADY the B operand: the accumulator layout of step 1
//              (4 consecutive rows per lane) is the k-run layout of the next MFMA; the implied k-slot permutation
//              (slots 0-3 <- hidden 4g..4g+3, slots 4-7 <- hidden 16+4g..) is baked into the packed W2 tile.
//   epilogue : out = x + gamma2 * (D2 + b2), optional second destination (ViT feature taps), and the row statistics
//              (mean, rstd) of the NEW x for the next block's LayerNorm, which the QKV GEMM applies on load.
// LDS row strides are == 2 (mod 4) sixteen-byte slots: conflict-free for the 16-lane ds_read_b128 service groups.
#include "common.h"
#include <cstdlib>

#ifndef MLP_WAVES_PER_SIMD
#define MLP_WAVES_PER_SIMD 2
#endif

#define TSTAMP(i) do {} while (0)

namespace {

#ifndef MLP_NW
#define MLP_NW 8
#endif
#ifndef MLP_TT384
#define MLP_TT384 1
#endif
constexpr int NW = MLP_NW, NTHR = NW * 64;      // waves / threads per workgroup (tuning builds: -DMLP_NW=4 -DMLP_WAVES_PER_SIMD=1)

// One LDS-DMA wave-instruction: 64 lanes x 16 bytes, lane l lands at lds_wave_base + 16 l; source = sbase + voff (bytes).
// Issued through inline assembly on purpose: hipcc treats the builtin form as a FLAT access that may touch LDS, and while
// one is pending every ds_read dependency is resolved with s_waitcnt lgkmcnt(0) - which serialises the fragment ring of
// the hidden loop. The copies are drained by the explicit vmcnt(0) in dma_sync(); hipcc's own vmcnt bookkeeping for
// ordinary loads stays conservative (the untracked pieces only make its counted waits wait longer).
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, const void* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff), "s"(sbase) : "memory");
}
// wave-uniform pointer -> SGPR pair (for dma16's scalar base when uniformity is not provable, e.g. derived from the wave id)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const uintptr_t v = (uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const void*)(((uintptr_t)hi << 32) | lo);
}
__device__ __forceinline__ void dma_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

struct MlpParams {
    void* x; long ldx;                 // in/out (M, C)
    const void* w1; const float* b1;   // (4C, C) with LN gamma folded in, b1' = b1 + W1 beta
    const void* w2p; const float* b2;  // chunk-major (4C/32, C, 32); (C)
    const float* gamma2;               // (C) LayerScale
    void* out2; long ld2;              // optional second destination
    float* stats_out;                  // optional (M, 2): mean, rstd of the updated rows (eps_next)
    long M; float eps, eps_next;
    int ntiles;                        // token tiles (16*TT rows each) in total; gridDim.x workgroups share them evenly
    // optional fused attention output projection: x <- x + gamma1 * (att Wp^T + bp) before the MLP
    const void* att; long ldatt; const void* wp; const float* bp; const float* gamma1;
    // optional chained LayerNorm + QKV projection of the NEXT block on the updated rows (vit.py:199, :123-130)
    const void* wqkv; const float* bqkv; void* q; void* k; void* vt; float qscale; int heads, hd, Tp;
    // FFN mode (decoder: linear1 -> ReLU -> linear2, models/transformer.py:507-512): workgroup (x, y) walks hidden chunks
    // [y * chunks_per_split, (y + 1) * chunks_per_split) of its tiles and writes the f32 partial products of linear2 to
    // partial[y][m][C]; no LayerNorm, no residual (lwdetr_ffn_finish sums the slabs)
    float* partial; int chunks_per_split;
    int wfrag;                         // few-token kernel: Wp / W1 / Wqkv are fragment-major (lwdetr_vit_block_few)
};

template <typename T, int C, int TT, bool PROJ, bool QKV, bool FFN = false>
__global__ __launch_bounds__(NTHR, MLP_WAVES_PER_SIMD) void mlp_kernel(const MlpParams p) {
    static_assert(!FFN || (!PROJ && !QKV), "FFN mode has no projection / chained QKV");
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    constexpr int KC = C / 32;                 // k-chunks of step 1
    constexpr int NT = C / 16;                 // output tiles of step 2
    constexpr int HID = 4 * C, NCH = HID / 32; // hidden chunks
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int W1_LD = C + 2 * EPC;         // +32 bytes: row stride == 2 (mod 4) slots for C = 192, 384, 768
    constexpr int W2_LD = 32 + 2 * EPC;        // +32 bytes: 6 slots (16-bit) / 10 slots (f32) per row, == 2 (mod 4)
    constexpr int W1_TILE = 32 * W1_LD, W2_TILE = C * W2_LD;
    constexpr int W1_TILE_PAD = (W1_TILE + 64 * EPC - 1) / (64 * EPC) * (64 * EPC);   // whole 1 KB DMA pieces
    constexpr int W2_TILE_PAD = (W2_TILE + 64 * EPC - 1) / (64 * EPC) * (64 * EPC);
    constexpr int TILE_STRIDE = W1_TILE_PAD + W2_TILE_PAD;
    // 16-bit C = 192 with the fused projection: the post-attention residual stream x1 of the workgroup's <= 7 tiles lives
    // in LDS from the projection to the epilogue (x is DMA'd in, updated in place, read back by the epilogue) instead of
    // going out to HBM and back: -39 MB of the kernel's 157 MB, no dependent global load in the projection loop, no store
    // for the chunk barriers to drain. Rows are C + 8 elements (400 B == 36 dwords mod 64: the 16 rows of a ds_read_b64 /
    // ds_write_b64 lane group start in 16 distinct even banks). The host deals at most X1_TILES tiles to a workgroup.
    constexpr bool X1LDS = PROJ && sizeof(T) == 2 && C == 192;
    constexpr int X1_LD = C + 8, X1_TILES = 7, X1_ELEMS = X1LDS ? X1_TILES * 16 * TT * X1_LD : 0;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = (T*)smem_raw;                    // [2][TILE_STRIDE] weight tiles, then fc1 bias (f32)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    TSTAMP(0);
    // this workgroup's contiguous run of tiles (sizes differ by at most one); waves beyond it run on clamped rows and
    // store nothing (they still take part in the weight staging and barriers)
    const int tb = p.ntiles / (int)gridDim.x, tr = p.ntiles % (int)gridDim.x;
    const int my_tiles = tb + ((int)blockIdx.x < tr ? 1 : 0);
    const int tile0 = (int)blockIdx.x * tb + ((int)blockIdx.x < tr ? (int)blockIdx.x : tr);
    const long m_wave = wave < my_tiles ? (long)(tile0 + wave) * (16 * TT) : p.M;
    T* __restrict__ X = (T*)p.x;
    const T* __restrict__ W1 = (const T*)p.w1;
    const T* __restrict__ W2 = (const T*)p.w2p;

    // ---- weight tile staging: asynchronous global -> LDS DMA (global_load_lds, 16 B per lane, no staging registers).
    // The LDS image keeps padded rows (conflict-free fragment reads); a wave-instruction fills 64 consecutive 16-byte
    // slots, lanes that fall on pad slots fetch a dummy (valid) address. The copies stay in flight across the whole
    // chunk body and are drained by the chunk-end barrier.
    constexpr int W1_SLOTS_ROW = W1_LD / EPC, W2_SLOTS_ROW = W2_LD / EPC;
    constexpr int W1_SLOTS = 32 * W1_SLOTS_ROW, W2_SLOTS = C * W2_SLOTS_ROW;
    constexpr int W1_INSTR = (W1_SLOTS + 63) / 64, W2_INSTR = (W2_SLOTS + 63) / 64;
    static_assert(W1_TILE % (64 * EPC) == 0 || true, "");
    // per-lane source offsets of every DMA piece this wave issues (chunk invariant; -1 = pad slot)
    constexpr int W1_MY = (W1_INSTR + NW - 1) / NW, W2_MY = (W2_INSTR + NW - 1) / NW;
    unsigned off1[W1_MY], off2[W2_MY];        // BYTE offsets (saddr + 32-bit voffset addressing: one VGPR per piece)
#pragma unroll
    for (int k = 0; k < W1_MY; ++k) {
        const int slot = (wave + NW * k) * 64 + lane, row = slot / W1_SLOTS_ROW, c = slot - row * W1_SLOTS_ROW;
        off1[k] = (row < 32 && c < C / EPC) ? (unsigned)((row * C + c * EPC) * sizeof(T)) : 0u;
    }
#pragma unroll
    for (int k = 0; k < W2_MY; ++k) {
        const int slot = (wave + NW * k) * 64 + lane, row = slot / W2_SLOTS_ROW, c = slot - row * W2_SLOTS_ROW;
        off2[k] = (row < C && c < 32 / EPC) ? (unsigned)((row * 32 + c * EPC) * sizeof(T)) : 0u;
    }
    auto stage_w = [&](int hc, int buf) {
        T* w1s = smem + buf * TILE_STRIDE;
        T* w2s = w1s + W1_TILE_PAD;
        const T* s1 = W1 + (long)hc * 32 * C;
        const T* s2 = W2 + (long)hc * 32 * C;
#pragma unroll
        for (int k = 0; k < W1_MY; ++k) {
            const int i = wave + NW * k;
            if (i < W1_INSTR)
                dma16(s1, off1[k], w1s + i * 64 * EPC);
        }
#pragma unroll
        for (int k = 0; k < W2_MY; ++k) {
            const int i = wave + NW * k;
            if (i < W2_INSTR)
                dma16(s2, off2[k], w2s + i * 64 * EPC);
        }
    };
    // a (32 rows x C) weight piece into the W1 slot of buffer `buf` (projection / QKV weights stream through it)
    auto stage_rows32 = [&](const T* src, int buf) {
        T* w1s = smem + buf * TILE_STRIDE;
#pragma unroll
        for (int k = 0; k < W1_MY; ++k) {
            const int i = wave + NW * k;
            if (i < W1_INSTR)
                dma16(src, off1[k], w1s + i * 64 * EPC);
        }
    };
    // fc1 bias (and the projection's bias / LayerScale) -> LDS once: ordinary global loads inside the loops would force
    // an early drain of the DMA queue
    constexpr int XCHG = QKV ? NW * 16 * TT * (C + 2 * EPC) : 0;
    T* x1s = smem + 2 * TILE_STRIDE + (wave < X1_TILES ? wave : 0) * 16 * TT * X1_LD;      // this wave's x1 rows (X1LDS)
    float* b1s = (float*)(smem + (2 * TILE_STRIDE + X1_ELEMS > XCHG ? 2 * TILE_STRIDE + X1_ELEMS : XCHG));
    // hidden chunks of this workgroup: all of them (ViT block) or the slice of its split (FFN mode; HID floats of LDS hold it)
    const int hc0 = FFN ? (int)blockIdx.y * p.chunks_per_split : 0;
    const int hc1 = FFN ? hc0 + p.chunks_per_split : NCH;
    if (FFN) { for (int i = tid; i < (hc1 - hc0) * 32; i += NTHR) b1s[i] = p.b1[hc0 * 32 + i]; }
    else { for (int i = tid; i < HID; i += NTHR) b1s[i] = p.b1[i]; }
    float* bps = b1s + HID;
    if (PROJ) for (int i = tid; i < C; i += NTHR) { bps[i] = p.bp[i]; bps[C + i] = p.gamma1[i]; }
    float* bqs = bps + 2 * C;
    if (QKV) for (int i = tid; i < 3 * C; i += NTHR) bqs[i] = p.bqkv[i];
    float* b2s = bqs + 3 * C;                    // fc2 bias and LayerScale (X1LDS: the epilogue issues no global load at all)
    if (X1LDS) for (int i = tid; i < C; i += NTHR) { b2s[i] = p.b2[i]; b2s[C + i] = p.gamma2[i]; }

    // ---- prologue: token rows -> B-operand fragments xf (lane: token l15, 8 channels per k-chunk)
    V8 xf[TT][KC];
    if (!PROJ) {
        stage_w(hc0, 0);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            long m = m_wave + t * 16 + l15; m = m < p.M ? m : p.M - 1;
            const T* xr = X + m * p.ldx + g * 8;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) xf[t][kc] = *(const V8*)(xr + kc * 32);
        }
    } else {
        // Fused attention output projection (reference vit.py:138, :206-216): x1 = x + gamma1 * (att Wp^T + bp), computed
        // as D[channel][token] so that the accumulators of two adjacent 16-channel tiles ARE the next MFMA's k-run
        // (slots 0-3 <- channels 32pc+4g.., slots 4-7 <- 32pc+16+4g..; fc1's columns are permuted to match on the host).
        // Wp streams through the W1 tile buffers in pieces of 32 output channels.
        const T* __restrict__ ATT = (const T*)p.att;
        const T* __restrict__ WP = (const T*)p.wp;
        stage_rows32(WP, 0);
        if (X1LDS && wave < X1_TILES) {
            // this wave's 16 TT rows of x -> x1s: 25 sixteen-byte slots per row (24 data + 1 pad), lane-linear pieces
            constexpr int SLOTS_ROW = X1_LD / 8, NSLOT = 16 * TT * SLOTS_ROW, NPIECE = (NSLOT + 63) / 64;
            const long mw = m_wave < p.M ? m_wave : (p.M - 16 * TT > 0 ? p.M - 16 * TT : 0);   // idle waves: any valid rows
            const void* xbase = uniform_ptr(X + mw * p.ldx);
#pragma unroll
            for (int k = 0; k < NPIECE; ++k) {
                const int slot = k * 64 + lane, row = slot / SLOTS_ROW, c = slot - row * SLOTS_ROW;
                long m = mw + row; m = m < p.M ? m : p.M - 1;
                const unsigned off = (row < 16 * TT && c < C / 8) ? (unsigned)(((m - mw) * p.ldx + c * 8) * (long)sizeof(T)) : 0u;
                if (slot < NSLOT) dma16(xbase, off, x1s + k * 64 * 8);      // last piece: only the lanes inside this wave's rows
            }
        }
        V8 af[TT][KC];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            long m = m_wave + t * 16 + l15; m = m < p.M ? m : p.M - 1;
            const T* ar = ATT + m * p.ldatt + g * 8;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) af[t][kc] = *(const V8*)(ar + kc * 32);
        }
        dma_sync();
#pragma unroll
        for (int pc = 0; pc < KC; ++pc) {
            const int buf = pc & 1;
            if (pc + 1 < KC) stage_rows32(WP + (long)(pc + 1) * 32 * C, buf ^ 1);
            else stage_w(0, buf ^ 1);                    // KC is even: the last piece sits in buffer 1, chunk 0 goes to 0
            const T* wps = smem + buf * TILE_STRIDE;
            f32x4 accp[2][TT];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < TT; ++t) accp[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const V8 a = *(const V8*)(wps + (h * 16 + l15) * W1_LD + kc * 32 + g * 8);
#pragma unroll
                    for (int t = 0; t < TT; ++t) accp[h][t] = Mma<T>::k32(a, af[t][kc], accp[h][t]);
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c0 = pc * 32 + h * 16 + g * 4;
                const f32x4 bb = *(const f32x4*)(bps + c0), gg = *(const f32x4*)(bps + C + c0);
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const long m = m_wave + t * 16 + l15;
                    const long mr = m < p.M ? m : p.M - 1;
                    V4 x1;
                    if (X1LDS) {
                        T* xs = x1s + (t * 16 + l15) * X1_LD + c0;
                        x1 = cvt4<T>(up4<T>(*(const V4*)xs) + gg * (accp[h][t] + bb));
                        if (wave < X1_TILES) *(V4*)xs = x1;          // wave 7 never owns a tile (and has no rows of its own)
                    } else {
                        x1 = cvt4<T>(up4<T>(*(const V4*)(X + mr * p.ldx + c0)) + gg * (accp[h][t] + bb));
                        if (m < p.M) *(V4*)(X + m * p.ldx + c0) = x1;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) xf[t][pc][h * 4 + e] = x1[e];
                }
            }
            dma_sync();
        }
    }
    TSTAMP(1);
    // ---- LayerNorm in registers (two-pass f32 statistics; the affine part is folded into W1 / b1 on the host)
#pragma unroll
    for (int t = 0; t < (FFN ? 0 : TT); ++t) {
        float s = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += to_f32<T>(xf[t][kc][e]);
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        const float mean = s * (1.f / C);
        float v = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dl = to_f32<T>(xf[t][kc][e]) - mean; v += dl * dl; }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[t][kc][e] = from_f32<T>((to_f32<T>(xf[t][kc][e]) - mean) * rstd);
    }

    f32x4 acc2[NT][TT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int t = 0; t < TT; ++t) acc2[n][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (!PROJ) dma_sync();
    TSTAMP(2);
    for (int hc = hc0; hc < hc1; ++hc) {
        const int buf = (hc - hc0) & 1;
        if (hc + 1 < hc1) stage_w(hc + 1, buf ^ 1);
        const T* w1s = smem + buf * TILE_STRIDE;
        const T* w2s = w1s + W1_TILE_PAD;
        // bias of this lane's 8 hidden units (rows 4g..4g+3 of the two 16-row tiles): the accumulators start from it
        const f32x4 bia0 = *(const f32x4*)(b1s + (hc - hc0) * 32 + g * 4), bia1 = *(const f32x4*)(b1s + (hc - hc0) * 32 + 16 + g * 4);
        // The 2*KC + NT weight fragments of the chunk are read through a ring RD deep: the read of fragment i + RD is
        // issued before the MFMAs of fragment i, so an LDS round trip is never exposed (left alone, hipcc issues every
        // ds_read right in front of its MFMAs and waits for it: ~100 exposed cycles per 34 cycles of MFMA).
        // MI355X measurements behind this loop (tools/ubench/issue.hip): plain VALU does NOT execute in the shadow of an
        // MFMA on the same SIMD, neither from the same wave nor from its partner (cycles add up: 17.5 per 16x16x32 MFMA
        // + 2.6 per VALU); transcendentals and ds_reads do. So the loop is as fast as its instruction count allows, and
        // software-pipelining GELU against the neighbouring chunks' MFMAs (tried: one stream, skewed halves, priorities)
        // buys nothing.
        constexpr int RD = 3, NF1 = 2 * KC, NF = NF1 + NT;
        auto frag = [&](int i) -> V8 {
            if (i < NF1) return *(const V8*)(w1s + ((i & 1) * 16 + l15) * W1_LD + (i >> 1) * 32 + g * 8);
            return *(const V8*)(w2s + ((i - NF1) * 16 + l15) * W2_LD + g * 8);     // k-slot order pre-permuted on the host
        };
        V8 fr[RD];
#pragma unroll
        for (int i = 0; i < RD; ++i) fr[i] = frag(i);
        // ---- step 1
        f32x4 acc1[2][TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) { acc1[0][t] = bia0; acc1[1][t] = bia1; }
#pragma unroll
        for (int i = 0; i < NF1; ++i) {
            const V8 a = fr[i % RD];
            fr[i % RD] = frag(i + RD);
#pragma unroll
            for (int t = 0; t < TT; ++t) acc1[i & 1][t] = Mma<T>::k32(a, xf[t][i >> 1], acc1[i & 1][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- GELU on the accumulator layout -> B operand of step 2
        V8 hf[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hf[t][e] = from_f32<T>(FFN ? fmaxf(acc1[0][t][e], 0.f) : gelu_for<T>(acc1[0][t][e]));
                hf[t][4 + e] = from_f32<T>(FFN ? fmaxf(acc1[1][t][e], 0.f) : gelu_for<T>(acc1[1][t][e]));
            }
        __builtin_amdgcn_sched_barrier(0);
        // ---- step 2
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int i = NF1 + n;
            const V8 a = fr[i % RD];
            if (i + RD < NF) fr[i % RD] = frag(i + RD);
#pragma unroll
            for (int t = 0; t < TT; ++t) acc2[n][t] = Mma<T>::k32(a, hf[t], acc2[n][t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        dma_sync();             // drains this chunk's DMA (vmcnt) and orders the buffer swap
    }

    TSTAMP(3);
    if (FFN) {      // partial linear2 products of this hidden slice: f32, lane holds channels n*16 + 4g .. +3 of token l15
        float* __restrict__ part = p.partial + (long)blockIdx.y * p.M * C;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const long m = m_wave + t * 16 + l15;
            if (m < p.M) {
#pragma unroll
                for (int n = 0; n < NT; ++n) *(f32x4*)(part + m * C + n * 16 + g * 4) = acc2[n][t];
            }
        }
        return;
    }
    // ---- epilogue: lane holds channels n*16 + 4g .. +3 of token l15
    T* __restrict__ O2 = (T*)p.out2;
    V8 xq[QKV ? TT : 1][QKV ? KC : 1];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const long m = m_wave + t * 16 + l15;
        const bool ok = m < p.M;
        const long mr = ok ? m : p.M - 1;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int c0 = n * 16 + g * 4;
            const f32x4 b2 = X1LDS ? *(const f32x4*)(b2s + c0) : *(const f32x4*)(p.b2 + c0);
            const f32x4 g2 = X1LDS ? *(const f32x4*)(b2s + C + c0) : *(const f32x4*)(p.gamma2 + c0);
            const f32x4 xr = up4<T>(X1LDS ? *(const V4*)(x1s + (t * 16 + l15) * X1_LD + c0) : *(const V4*)(X + mr * p.ldx + c0));
            // round to the storage type now: the statistics below describe exactly what the next LayerNorm reads
            acc2[n][t] = up4<T>(cvt4<T>(xr + g2 * (acc2[n][t] + b2)));
        }
        if (ok) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const V4 o = cvt4<T>(acc2[n][t]);
                *(V4*)(X + m * p.ldx + n * 16 + g * 4) = o;
                if (O2) *(V4*)(O2 + m * p.ld2 + n * 16 + g * 4) = o;
            }
        }
        if (p.stats_out || QKV) {
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < NT; ++n) s += acc2[n][t][0] + acc2[n][t][1] + acc2[n][t][2] + acc2[n][t][3];
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            const float mean = s * (1.f / C);
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dl = acc2[n][t][e] - mean; v += dl * dl; }
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps_next);
            if (p.stats_out && ok && g == 0) { p.stats_out[2 * m] = mean; p.stats_out[2 * m + 1] = rstd; }
            if (QKV) {
                // normalised rows as B-operand fragments, in accumulator (k-slot) order: the QKV weight's columns carry
                // the same permutation inside every 32-chunk
#pragma unroll
                for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xq[t][kc][e] = from_f32<T>((acc2[2 * kc][t][e] - mean) * rstd);
                        xq[t][kc][4 + e] = from_f32<T>((acc2[2 * kc + 1][t][e] - mean) * rstd);
                    }
            }
        }
    }
    TSTAMP(4);
    if (QKV) {
        // ---- chained LayerNorm + QKV of the next block, re-partitioned: the normalised rows of ALL tiles of the
        // workgroup are exchanged through LDS, and each wave keeps ITS share of the QKV weight (every 8th 16-feature
        // tile, A-stationary in registers, fetched once straight from L2) and sweeps the workgroup's tokens. No weight
        // streaming, no per-piece DMA latency chain, one barrier.
        // Q, K: D[feature][token] (lane: 4 consecutive features of a head, token l15) -> (B, heads, Tp, hd);
        // V   : operands swapped, D[token][feature] (lane: 4 consecutive tokens, feature l15) -> V^T (B, heads, hd, Tp).
        constexpr int XLD = C + 2 * EPC;                  // row stride == 2 (mod 4) slots: conflict-free fragment reads
        constexpr int NTQ = 3 * C / 16, NJ = (NTQ + NW - 1) / NW;
        const T* __restrict__ WQ = (const T*)p.wqkv;
        T* __restrict__ Qo = (T*)p.q; T* __restrict__ Ko = (T*)p.k; T* __restrict__ Vo = (T*)p.vt;
        if (X1LDS) __syncthreads();      // the exchange area overlays the x1 rows: every wave is done reading its own
        constexpr int FRAG_REGS = KC * (int)sizeof(V8) / 4;
        constexpr int JP = (120 / FRAG_REGS) < 1 ? 1 : ((120 / FRAG_REGS) > NJ ? NJ : (120 / FRAG_REGS));   // tiles per pass
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                *(V8*)(smem + ((wave * TT + t) * 16 + l15) * XLD + kc * 32 + g * 8) = xq[t][kc];
        __syncthreads();
        TSTAMP(5);
        const long m_blk = (long)tile0 * (16 * TT);
        const int ntt = my_tiles * TT;                    // 16-token tiles that carry real rows
#pragma unroll 1
        for (int j0 = 0; j0 < NJ; j0 += JP) {
            V8 wq[JP][KC];
#pragma unroll
            for (int j = 0; j < JP; ++j) {
                const int nt = wave + NW * (j0 + j);
                const T* wr = WQ + (long)((nt < NTQ ? nt : NTQ - 1) * 16 + l15) * C + g * 8;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) wq[j][kc] = *(const V8*)(wr + kc * 32);
            }
            // destination offsets split into a per-tile-column part (here, once per pass) and a per-token part (once per
            // token tile): no integer division inside the MFMA loop
            long coff[JP]; float4 cbias[JP]; int csg[JP];
#pragma unroll
            for (int j = 0; j < JP; ++j) {
                const int nt = wave + NW * (j0 + j);
                const int ntc = nt < NTQ ? nt : NTQ - 1;
                const int sg = ntc / (C / 16), nl0 = (ntc - sg * (C / 16)) * 16;
                csg[j] = nt < NTQ ? sg : -1;
                if (sg < 2) {
                    const int nl = nl0 + g * 4, hh = nl / p.hd, dd = nl - hh * p.hd;
                    coff[j] = (long)hh * p.Tp * p.hd + dd;
                    const f32x4 bb = *(const f32x4*)(bqs + sg * C + nl);
                    cbias[j] = make_float4(bb[0], bb[1], bb[2], bb[3]);
                } else {
                    const int nl = nl0 + l15, hh = nl / p.hd, dd = nl - hh * p.hd;
                    coff[j] = ((long)hh * p.hd + dd) * p.Tp;
                    const float bb = bqs[2 * C + nl];
                    cbias[j] = make_float4(bb, bb, bb, bb);
                }
            }
#pragma unroll 1
            for (int tt = 0; tt < ntt; ++tt) {
                V8 xr[KC];
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) xr[kc] = *(const V8*)(smem + (tt * 16 + l15) * XLD + kc * 32 + g * 8);
                const int mq = (int)m_blk + tt * 16 + l15, mv = (int)m_blk + tt * 16 + g * 4;
                const int bq_ = mq / p.Tp, bv_ = mv / p.Tp;
                const long roff_qk = ((long)bq_ * p.heads * p.Tp + (mq - bq_ * p.Tp)) * p.hd;
                const long roff_v = (long)bv_ * p.heads * p.hd * p.Tp + (mv - bv_ * p.Tp);
                const bool okq = mq < p.M, okv = mv < p.M;
#pragma unroll
                for (int j = 0; j < JP; ++j) {
                    if (csg[j] < 0) continue;
                    f32x4 acc = {cbias[j].x, cbias[j].y, cbias[j].z, cbias[j].w};
                    if (csg[j] < 2) {
#pragma unroll
                        for (int kc = 0; kc < KC; ++kc) acc = Mma<T>::k32(wq[j][kc], xr[kc], acc);
                        if (okq) {
                            T* dst = csg[j] == 0 ? Qo : Ko;
                            *(V4*)(dst + roff_qk + coff[j]) = cvt4<T>(acc * (csg[j] == 0 ? p.qscale : 1.f));
                        }
                    } else {
#pragma unroll
                        for (int kc = 0; kc < KC; ++kc) acc = Mma<T>::k32(xr[kc], wq[j][kc], acc);
                        if (okv) *(V4*)(Vo + roff_v + coff[j]) = cvt4<T>(acc);   // M, Tp multiples of 4: whole 4-token run
                    }
                }
            }
        }
    }
    TSTAMP(6);
}

// ---- few-token variant (16-bit, C = 192, with the projection; a handful of images: the single-image latency path).
// mlp_kernel gives a wave one tile and the whole hidden dimension: at 50 tiles that is 50 busy waves on the chip, each
// walking 24 chunks alone (~70 us). Here ONE 32-token tile is a workgroup and its 8 waves split the work the other way:
//   projection : waves 0-5 each produce 32 output channels of x1 (weights straight from L2 into A fragments) -> LDS
//   LayerNorm  : every wave reads the whole x1 tile back as its B fragments (redundant, ~100 VALU)
//   hidden     : wave w takes hidden chunks w, w + 8, w + 16 (3 of 24), weights straight from L2, partial fc2 sums
//   reduction  : the 8 partial tiles are summed through LDS in two rounds of 6 channel tiles; waves 0-5 finish one channel
//                tile each (bias, LayerScale, residual, stores) and put the new rows back into LDS
//   chained QKV: every wave normalises the whole tile again and computes its share of the 36 feature tiles from registers.
// No weight goes through LDS, no DMA ring, 6 barriers in all. Same packed weights and the same arithmetic per element as
// the large kernel, except for the order in which the 24 partial sums of fc2 are added (f32).
// TT = 16-token tiles per workgroup. 2 (32 tokens) from a few images up; 1 for one or two images: twice the workgroups (100
// for one 640 x 640 image on 256 CUs) and, with half the accumulators, room to keep the fc2 fragments of a chunk and the
// fc1 fragments of the wave's NEXT chunk in flight while the current one multiplies - the kernel is a chain of L2 round
// trips for weights (two per hidden chunk, one per QKV feature tile), and at this size nothing else matters.
template <typename T, bool QKV, int TT, bool FRAG = false>
__global__ __launch_bounds__(NTHR, 1) void mlp_small_kernel(const MlpParams p) {
    constexpr int C = 192, KC = C / 32, NT = C / 16, HID = 4 * C, X1_LD = C + 8;
    // FRAG: Wp, W1 and Wqkv arrive fragment-major - [row tile of 16][k-chunk of 32][16][32], every 16 x 32 MFMA A fragment one contiguous KB
    // (lwdetr_amd.kernels.pack_frag16) - instead of row-major: a fragment load is then 8 whole 128-byte lines instead of 16 half lines 384 bytes apart
    auto frag = [&](const T* w, int rt, int kc, int l15_, int g_) { return FRAG ? w + ((long)(rt * KC + kc) * 16 + l15_) * 32 + g_ * 8 : w + (long)(rt * 16 + l15_) * C + kc * 32 + g_ * 8; };
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    static_assert(sizeof(T) == 2, "");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* x1s = (T*)smem_raw;                                        // [16 TT][X1_LD]: x1, then the block's output rows
    float* part = (float*)(x1s + 16 * TT * X1_LD);                // [8 waves][6 tiles][TT][64 lanes][4]
    float* b1s = part + NW * 6 * TT * 256;
    float* bps = b1s + HID; float* bqs = bps + 2 * C; float* b2s = bqs + 3 * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const long m0 = (long)blockIdx.x * (16 * TT);
    T* __restrict__ X = (T*)p.x;
    const T* __restrict__ W1 = (const T*)p.w1;
    const T* __restrict__ W2 = (const T*)p.w2p;
    for (int i = tid; i < HID; i += NTHR) b1s[i] = p.b1[i];
    for (int i = tid; i < C; i += NTHR) { bps[i] = p.bp[i]; bps[C + i] = p.gamma1[i]; b2s[i] = p.b2[i]; b2s[C + i] = p.gamma2[i]; }
    if (QKV) for (int i = tid; i < 3 * C; i += NTHR) bqs[i] = p.bqkv[i];
    long mrow[TT]; bool mok[TT];                                  // this lane's token rows (clamped for loads)
#pragma unroll
    for (int t = 0; t < TT; ++t) { const long m = m0 + t * 16 + l15; mok[t] = m < p.M; mrow[t] = mok[t] ? m : p.M - 1; }
    __syncthreads();                                              // biases visible

    // ---- projection: wave pc < 6 -> channels 32 pc .. 32 pc + 31 of x1 = x + gamma1 * (att Wp^T + bp)
    if (wave < KC) {
        const int pc = wave;
        const T* __restrict__ ATT = (const T*)p.att;
        const T* __restrict__ WP = (const T*)p.wp;
        V8 af[TT][KC], wa[2][KC];
        V4 xr[2][TT];
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) af[t][kc] = *(const V8*)(ATT + mrow[t] * p.ldatt + kc * 32 + g * 8);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) wa[h][kc] = *(const V8*)frag(WP, pc * 2 + h, kc, l15, g);
#pragma unroll
            for (int t = 0; t < TT; ++t) xr[h][t] = *(const V4*)(X + mrow[t] * p.ldx + pc * 32 + h * 16 + g * 4);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c0 = pc * 32 + h * 16 + g * 4;
            const f32x4 bb = *(const f32x4*)(bps + c0), gg = *(const f32x4*)(bps + C + c0);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) a = Mma<T>::k32(wa[h][kc], af[t][kc], a);
                *(V4*)(x1s + (t * 16 + l15) * X1_LD + c0) = cvt4<T>(up4<T>(xr[h][t]) + gg * (a + bb));
            }
        }
    }
    __syncthreads();

    // ---- every wave: the whole x1 tile as B fragments (k-slot order of the projection's accumulators), LayerNorm
    V8 xf[TT][KC];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const V4 v = *(const V4*)(x1s + (t * 16 + l15) * X1_LD + kc * 32 + h * 16 + g * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) xf[t][kc][h * 4 + e] = v[e];
            }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float sm = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += to_f32<T>(xf[t][kc][e]);
        sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
        const float mean = sm * (1.f / C);
        float v = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dl = to_f32<T>(xf[t][kc][e]) - mean; v += dl * dl; }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) xf[t][kc][e] = from_f32<T>((to_f32<T>(xf[t][kc][e]) - mean) * rstd);
    }

    // ---- hidden chunks wave, wave + 8, wave + 16
    f32x4 acc2[NT][TT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int t = 0; t < TT; ++t) acc2[n][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (TT == 1) {
        // fc1 fragments of the wave's next chunk are requested as soon as the current ones have been multiplied, the fc2
        // fragments of a chunk before its fc1 MFMAs: one L2 round trip per chunk is exposed instead of two
        V8 f1[2 * KC];
#pragma unroll
        for (int i = 0; i < 2 * KC; ++i) f1[i] = *(const V8*)frag(W1, wave * 2 + (i & 1), i >> 1, l15, g);
#pragma unroll 1
        for (int hc = wave; hc < HID / 32; hc += NW) {
            V8 f2[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) f2[n] = *(const V8*)(W2 + (long)hc * 32 * C + (n * 16 + l15) * 32 + g * 8);
            const f32x4 bia0 = *(const f32x4*)(b1s + hc * 32 + g * 4), bia1 = *(const f32x4*)(b1s + hc * 32 + 16 + g * 4);
            f32x4 acc1[2] = {bia0, bia1};
#pragma unroll
            for (int i = 0; i < 2 * KC; ++i) acc1[i & 1] = Mma<T>::k32(f1[i], xf[0][i >> 1], acc1[i & 1]);
            const int hn = hc + NW < HID / 32 ? hc + NW : hc;            // last chunk: a harmless reload
#pragma unroll
            for (int i = 0; i < 2 * KC; ++i) f1[i] = *(const V8*)frag(W1, hn * 2 + (i & 1), i >> 1, l15, g);
            V8 hf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hf[e] = from_f32<T>(gelu_for<T>(acc1[0][e]));
                hf[4 + e] = from_f32<T>(gelu_for<T>(acc1[1][e]));
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) acc2[n][0] = Mma<T>::k32(f2[n], hf, acc2[n][0]);
        }
    } else {
#pragma unroll 1
        for (int hc = wave; hc < HID / 32; hc += NW) {
            V8 fr[2 * KC];
    #pragma unroll
            for (int i = 0; i < 2 * KC; ++i)        // fc1 fragment i = (kc = i / 2, h = i % 2)
                fr[i] = *(const V8*)frag(W1, hc * 2 + (i & 1), i >> 1, l15, g);
            const f32x4 bia0 = *(const f32x4*)(b1s + hc * 32 + g * 4), bia1 = *(const f32x4*)(b1s + hc * 32 + 16 + g * 4);
            f32x4 acc1[2][TT];
    #pragma unroll
            for (int t = 0; t < TT; ++t) { acc1[0][t] = bia0; acc1[1][t] = bia1; }
    #pragma unroll
            for (int i = 0; i < 2 * KC; ++i)
    #pragma unroll
                for (int t = 0; t < TT; ++t) acc1[i & 1][t] = Mma<T>::k32(fr[i], xf[t][i >> 1], acc1[i & 1][t]);
    #pragma unroll
            for (int n = 0; n < NT; ++n)            // fc2 fragments of the chunk: in flight while GELU runs
                fr[n] = *(const V8*)(W2 + (long)hc * 32 * C + (n * 16 + l15) * 32 + g * 8);
            V8 hf[TT];
    #pragma unroll
            for (int t = 0; t < TT; ++t)
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hf[t][e] = from_f32<T>(gelu_for<T>(acc1[0][t][e]));
                    hf[t][4 + e] = from_f32<T>(gelu_for<T>(acc1[1][t][e]));
                }
    #pragma unroll
            for (int n = 0; n < NT; ++n)
    #pragma unroll
                for (int t = 0; t < TT; ++t) acc2[n][t] = Mma<T>::k32(fr[n], hf[t], acc2[n][t]);
        }
    
}

    // ---- reduction over the 8 waves + epilogue, 6 channel tiles per round; wave w < 6 finishes channel tile 6 r + w
    T* __restrict__ O2 = (T*)p.out2;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int n = 0; n < 6; ++n)
#pragma unroll
            for (int t = 0; t < TT; ++t) *(f32x4*)(part + ((wave * 6 + n) * TT + t) * 256 + lane * 4) = acc2[6 * r + n][t];
        __syncthreads();
        if (wave < 6) {
            const int c0 = (6 * r + wave) * 16 + g * 4;
            const f32x4 b2 = *(const f32x4*)(b2s + c0), g2 = *(const f32x4*)(b2s + C + c0);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                f32x4 sum = *(const f32x4*)(part + ((0 * 6 + wave) * TT + t) * 256 + lane * 4);
#pragma unroll
                for (int w = 1; w < NW; ++w) sum += *(const f32x4*)(part + ((w * 6 + wave) * TT + t) * 256 + lane * 4);
                T* xs = x1s + (t * 16 + l15) * X1_LD + c0;
                const V4 o = cvt4<T>(up4<T>(*(const V4*)xs) + g2 * (sum + b2));
                *(V4*)xs = o;
                if (mok[t]) {
                    *(V4*)(X + mrow[t] * p.ldx + c0) = o;
                    if (O2) *(V4*)(O2 + mrow[t] * p.ld2 + c0) = o;
                }
            }
        }
        __syncthreads();
    }

    if (!(p.stats_out || QKV)) return;
    // ---- statistics of the new rows (every wave, from LDS) and the chained LayerNorm + QKV of the next block
    V8 xq[TT][KC];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float xv[KC][8];
        float sm = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const V4 v = *(const V4*)(x1s + (t * 16 + l15) * X1_LD + kc * 32 + h * 16 + g * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { xv[kc][h * 4 + e] = to_f32<T>(v[e]); sm += xv[kc][h * 4 + e]; }
            }
        sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
        const float mean = sm * (1.f / C);
        float v = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dl = xv[kc][e] - mean; v += dl * dl; }
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        const float rstd = 1.f / sqrtf(v * (1.f / C) + p.eps_next);
        if (p.stats_out && wave == 0 && mok[t] && g == 0) { p.stats_out[2 * mrow[t]] = mean; p.stats_out[2 * mrow[t] + 1] = rstd; }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int e = 0; e < 8; ++e) xq[t][kc][e] = from_f32<T>((xv[kc][e] - mean) * rstd);
    }
    if (!QKV) return;
    // Q, K: D[feature][token] -> (B, heads, Tp, hd); V: operands swapped, D[token][feature] -> V^T (B, heads, hd, Tp)
    constexpr int NTQ = 3 * C / 16;
    const T* __restrict__ WQ = (const T*)p.wqkv;
    T* __restrict__ Qo = (T*)p.q; T* __restrict__ Ko = (T*)p.k; T* __restrict__ Vo = (T*)p.vt;
    // feature tiles wave, wave + 8, ...: the weights of the next tile are requested before the current one is multiplied
    constexpr int NIT = (NTQ + NW - 1) / NW;
    V8 wq[2][KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) wq[0][kc] = *(const V8*)frag(WQ, wave, kc, l15, g);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int nt = wave + it * NW;
        if (it + 1 < NIT) {
            const int ntn = nt + NW < NTQ ? nt + NW : NTQ - 1;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) wq[(it + 1) & 1][kc] = *(const V8*)frag(WQ, ntn, kc, l15, g);
        }
        if (nt >= NTQ) break;                                          // wave-uniform
        const int sg = nt / (C / 16), nl0 = (nt - sg * (C / 16)) * 16;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if (sg < 2) {
                const int nl = nl0 + g * 4, hh = nl / p.hd, dd = nl - hh * p.hd;
                f32x4 acc = *(const f32x4*)(bqs + sg * C + nl);
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) acc = Mma<T>::k32(wq[it & 1][kc], xq[t][kc], acc);
                const int mq = (int)m0 + t * 16 + l15, bq_ = mq / p.Tp;
                if (mq < p.M) {
                    T* dst = (sg == 0 ? Qo : Ko) + ((long)bq_ * p.heads * p.Tp + (mq - bq_ * p.Tp)) * p.hd + (long)hh * p.Tp * p.hd + dd;
                    *(V4*)dst = cvt4<T>(acc * (sg == 0 ? p.qscale : 1.f));
                }
            } else {
                const int nl = nl0 + l15, hh = nl / p.hd, dd = nl - hh * p.hd;
                const float bb = bqs[2 * C + nl];
                f32x4 acc = {bb, bb, bb, bb};
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) acc = Mma<T>::k32(xq[t][kc], wq[it & 1][kc], acc);
                const int mv = (int)m0 + t * 16 + g * 4, bv_ = mv / p.Tp;
                if (mv < p.M)
                    *(V4*)(Vo + (long)bv_ * p.heads * p.hd * p.Tp + (mv - bv_ * p.Tp) + ((long)hh * p.hd + dd) * p.Tp) = cvt4<T>(acc);
            }
        }
    }
}
constexpr long MLP_SMALL_TT1_MAX_ROWS = 3200;      // one or two 640 x 640 images: 16-token workgroups (see mlp_small_kernel)
template <typename T, bool QKV, int TT, bool FRAG = false>
int launch_mlp_small_tt(const MlpParams& p, hipStream_t st) {
    constexpr int C = 192;
    constexpr size_t lds = 16 * TT * (C + 8) * sizeof(T) + (size_t)NW * 6 * TT * 256 * sizeof(float) + 11 * C * sizeof(float);
    static bool attr_done[16] = {};            // per device (a process may drive several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    if (!attr_done[dev]) {
        if (hipFuncSetAttribute((const void*)mlp_small_kernel<T, QKV, TT, FRAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        attr_done[dev] = true;
    }
    const long blocks = (p.M + 16 * TT - 1) / (16 * TT);
    ProfScope ps(KID_MLP, (16.0 + 2.0 + (QKV ? 6.0 : 0.0)) * p.M * C * C, (double)p.M * C * sizeof(T) * 3 + (QKV ? 3.0 : 0.0) * p.M * C * sizeof(T), st);
    hipLaunchKernelGGL((mlp_small_kernel<T, QKV, TT, FRAG>), dim3((unsigned)blocks), dim3(NTHR), lds, st, p);
    return lwdetr_check_launch();
}
template <typename T, bool QKV>
int launch_mlp_small(const MlpParams& p, hipStream_t st) {
    const int tt = (int)lwdetr_knob(KNOB_MLP_SMALL_TT, p.M <= MLP_SMALL_TT1_MAX_ROWS ? 1 : 2);          // tuning: 1 / 2 = token tiles per workgroup
    if (p.wfrag) return tt == 1 ? launch_mlp_small_tt<T, QKV, 1, true>(p, st) : launch_mlp_small_tt<T, QKV, 2, true>(p, st);
    return tt == 1 ? launch_mlp_small_tt<T, QKV, 1>(p, st) : launch_mlp_small_tt<T, QKV, 2>(p, st);
}

template <typename T, int C, int TT, bool PROJ, bool QKV>
int launch_mlp_p(const MlpParams& p, hipStream_t st) {
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int PIECE = 64 * EPC;
    constexpr int W1P = (32 * (C + 2 * EPC) + PIECE - 1) / PIECE * PIECE, W2P = (C * (32 + 2 * EPC) + PIECE - 1) / PIECE * PIECE;
    constexpr size_t tiles_b = 2 * (size_t)(W1P + W2P) * sizeof(T);
    constexpr size_t xchg_b = QKV ? (size_t)NW * 16 * TT * (C + 2 * EPC) * sizeof(T) : 0;
    constexpr bool X1LDS = PROJ && sizeof(T) == 2 && C == 192;
    constexpr int X1_TILES = 7;
    constexpr size_t x1_b = X1LDS ? (size_t)X1_TILES * 16 * TT * (C + 8) * sizeof(T) : 0;
    constexpr size_t lds = (tiles_b + x1_b > xchg_b ? tiles_b + x1_b : xchg_b) + (9 + (X1LDS ? 2 : 0)) * C * sizeof(float);
    static int ncu_dev[16] = {};               // per device: attribute set + CU count (0 = not initialised)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    if (!ncu_dev[dev]) {
        hipDeviceProp_t prop;
        if (hipFuncSetAttribute((const void*)mlp_kernel<T, C, TT, PROJ, QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        ncu_dev[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int ncu = ncu_dev[dev];
    MlpParams q = p;
    q.ntiles = (int)((p.M + 16 * TT - 1) / (16 * TT));
    long blocks = q.ntiles < ncu ? q.ntiles : ncu;                       // one workgroup per CU, tiles dealt evenly
    constexpr int TPW = X1LDS ? X1_TILES : NW;                          // tiles a workgroup can take (X1LDS: LDS rows for 7)
    if ((long)q.ntiles > blocks * TPW) blocks = (q.ntiles + TPW - 1) / TPW;   // more than that per CU: extra rounds
    ProfScope ps(KID_MLP, (16.0 + (PROJ ? 2.0 : 0.0) + (QKV ? 6.0 : 0.0)) * p.M * C * C,
                 (double)p.M * C * sizeof(T) * (PROJ ? 3 : 2) + (QKV ? 3.0 : 0.0) * p.M * C * sizeof(T), st);
    hipLaunchKernelGGL((mlp_kernel<T, C, TT, PROJ, QKV>), dim3((unsigned)blocks), dim3(NTHR), lds, st, q);
    return lwdetr_check_launch();
}

// FFN mode launch: full workgroups (8 tiles each) x S hidden splits, S = the largest power of two that keeps the grid within
// one workgroup per CU (the weight tiles take the LDS of a CU) and leaves every split at least two chunks (double buffer).
constexpr int FFN_MAX_SPLITS = 16;   // measured (tools/ffn_bench.py, M = 300): 32 -> 22.3 us, 16 -> 20.0, 8 -> 22.1, 4 -> 30.4
template <typename T, int C, int TT>
int launch_ffn(const MlpParams& p0, int hid, hipStream_t st, int* splits_out) {
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int PIECE = 64 * EPC;
    constexpr int W1P = (32 * (C + 2 * EPC) + PIECE - 1) / PIECE * PIECE, W2P = (C * (32 + 2 * EPC) + PIECE - 1) / PIECE * PIECE;
    constexpr size_t lds = 2 * (size_t)(W1P + W2P) * sizeof(T) + 9 * C * sizeof(float);
    static int ncu_dev[16] = {};               // per device: attribute set + CU count (0 = not initialised)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_LAUNCH;
    if (!ncu_dev[dev]) {
        hipDeviceProp_t prop;
        if (hipFuncSetAttribute((const void*)mlp_kernel<T, C, TT, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return LWDETR_ERR_LAUNCH;
        ncu_dev[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int ncu = ncu_dev[dev];
    MlpParams q = p0;
    q.ntiles = (int)((p0.M + 16 * TT - 1) / (16 * TT));
    int blocks = (q.ntiles + NW - 1) / NW;
    const int nch = hid / 32;
    int S = 1;
    while ((long)(nch / S) * 32 > 4 * C && nch % (S * 2) == 0) S *= 2;       // the bias slice lives in the 4C floats of LDS
    if ((long)(nch / S) * 32 > 4 * C) return LWDETR_ERR_UNSUPPORTED;
    const int s_max = (int)lwdetr_knob(KNOB_FFN_SPLITS, FFN_MAX_SPLITS);   // tuning
    while (S * 2 * blocks <= ncu && nch % (S * 2) == 0 && nch / (S * 2) >= 2 && S * 2 <= s_max) S *= 2;
    if (ncu / S > blocks) blocks = ncu / S < q.ntiles ? ncu / S : q.ntiles;   // spare CUs: fewer tiles per workgroup
    q.chunks_per_split = nch / S;
    if (splits_out) { *splits_out = S; return LWDETR_OK; }
    ProfScope ps(KID_MLP, 4.0 * p0.M * C * hid, (double)p0.M * C * (sizeof(T) + 4.0 * S) + 2.0 * C * hid * sizeof(T), st);
    hipLaunchKernelGGL((mlp_kernel<T, C, TT, false, false, true>), dim3((unsigned)blocks, (unsigned)S), dim3(NTHR), lds, st, q);
    return lwdetr_check_launch();
}

constexpr long MLP_SMALL_MAX_ROWS = 12800;       // below ~8 images the tile-per-workgroup kernel wins (see mlp_small_kernel)

template <typename T, int C, int TT>
int launch_mlp(const MlpParams& p, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && C == 192) {
        const bool small_ok = p.att && (lwdetr_knob_is_set(KNOB_MLP_SMALL) ? lwdetr_knob(KNOB_MLP_SMALL, 0) == 1 : p.M < MLP_SMALL_MAX_ROWS);   // tuning: 0 = never, 1 = always
        if (small_ok) return p.wqkv ? launch_mlp_small<T, true>(p, st) : launch_mlp_small<T, false>(p, st);
    }
    if (p.att && p.wqkv) return launch_mlp_p<T, C, TT, true, true>(p, st);
    if (p.att) return launch_mlp_p<T, C, TT, true, false>(p, st);
    if (p.wqkv) return LWDETR_ERR_UNSUPPORTED;          // the chained QKV is only built together with the projection
    return launch_mlp_p<T, C, TT, false, false>(p, st);
}

template <typename T, int TT16, int TT32>
int dispatch_c(const MlpParams& p, int C, hipStream_t st) {
    switch (C) {
        case 192: return launch_mlp<T, 192, TT16>(p, st);
        case 384: return launch_mlp<T, 384, TT32>(p, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" int lwdetr_mlp_fused(void* x, long ldx, const void* w1_folded, const float* b1_folded, const void* w2_chunked,
                                const float* b2, const float* gamma2, void* out2, long ld2, float* stats_out, long M,
                                int C, float eps, float eps_next, const void* att, long ldatt, const void* wp,
                                const float* bp, const float* gamma1, const void* wqkv_next, const float* bqkv_next,
                                void* q_out, void* k_out, void* vt_out, float qscale, int heads, int hd, int Tp,
                                int dtype, void* hip_stream) {
    if (!x || !w1_folded || !b1_folded || !w2_chunked || !b2 || !gamma2 || M < 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    if (ldx % 8 != 0 || (out2 && ld2 % 8 != 0)) return LWDETR_ERR_BAD_ARG;
    if (att && (!wp || !bp || !gamma1 || ldatt % 8 != 0)) return LWDETR_ERR_BAD_ARG;
    MlpParams p;
    p.wfrag = 0;
    p.att = att; p.ldatt = ldatt; p.wp = wp; p.bp = bp; p.gamma1 = gamma1;
    p.wqkv = wqkv_next; p.bqkv = bqkv_next; p.q = q_out; p.k = k_out; p.vt = vt_out; p.qscale = qscale;
    p.heads = heads; p.hd = hd; p.Tp = Tp;
    if (wqkv_next && (!bqkv_next || !q_out || !k_out || !vt_out || heads <= 0 || hd % 4 != 0 || heads * hd != C || Tp % 4 != 0 ||
                      M % 4 != 0))
        return LWDETR_ERR_BAD_ARG;
    p.x = x; p.ldx = ldx; p.w1 = w1_folded; p.b1 = b1_folded; p.w2p = w2_chunked; p.b2 = b2; p.gamma2 = gamma2;
    p.out2 = out2; p.ld2 = ld2; p.stats_out = stats_out; p.M = M; p.eps = eps; p.eps_next = eps_next;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F16: return dispatch_c<f16, 2, MLP_TT384>(p, C, st);
        case DT_BF16: return dispatch_c<bf16, 2, MLP_TT384>(p, C, st);
        case DT_F32: return dispatch_c<float, 1, 1>(p, C, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

// The few-token form on its own entry point, with FRAGMENT-MAJOR weights (round 6). mlp_small_kernel streams every weight straight from L2 into
// MFMA A fragments (16 rows x 32 k per wave-load); out of a row-major (R, 192) matrix such a load is 16 segments of 64 bytes, 384 bytes apart -
// half lines - and the kernel's time WAS that load path: with Wp / W1 / Wqkv re-laid out so that every fragment is one contiguous KB
// (lwdetr_amd.kernels.pack_frag16: [R / 16][K / 32][16][32]; W2 is chunk-major already) a block launch at one 640 x 640 image goes 30 -> 22 us,
// the single-image forward 0.857 -> 0.786 ms (profiles/r6b_*). Same arithmetic, bit-identical results.
extern "C" int lwdetr_vit_block_few(void* x, long ldx, const void* w1_frag, const float* b1_folded, const void* w2_chunked, const float* b2,
                                    const float* gamma2, void* out2, long ld2, float* stats_out, long M, int C, float eps, float eps_next,
                                    const void* att, long ldatt, const void* wp_frag, const float* bp, const float* gamma1,
                                    const void* wqkv_frag_next, const float* bqkv_next, void* q_out, void* k_out, void* vt_out, float qscale,
                                    int heads, int hd, int Tp, int dtype, void* hip_stream) {
    if (!x || !w1_frag || !b1_folded || !w2_chunked || !b2 || !gamma2 || !att || !wp_frag || !bp || !gamma1 || M < 0) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    if (C != 192 || (dtype != DT_F16 && dtype != DT_BF16) || M >= MLP_SMALL_MAX_ROWS) return LWDETR_ERR_UNSUPPORTED;
    if (ldx % 8 != 0 || (out2 && ld2 % 8 != 0) || ldatt % 8 != 0) return LWDETR_ERR_BAD_ARG;
    if (wqkv_frag_next && (!bqkv_next || !q_out || !k_out || !vt_out || heads <= 0 || hd % 4 != 0 || heads * hd != C || Tp % 4 != 0 || M % 4 != 0))
        return LWDETR_ERR_BAD_ARG;
    MlpParams p;
    p.wfrag = 1;
    p.att = att; p.ldatt = ldatt; p.wp = wp_frag; p.bp = bp; p.gamma1 = gamma1;
    p.wqkv = wqkv_frag_next; p.bqkv = bqkv_next; p.q = q_out; p.k = k_out; p.vt = vt_out; p.qscale = qscale;
    p.heads = heads; p.hd = hd; p.Tp = Tp;
    p.x = x; p.ldx = ldx; p.w1 = w1_frag; p.b1 = b1_folded; p.w2p = w2_chunked; p.b2 = b2; p.gamma2 = gamma2;
    p.out2 = out2; p.ld2 = ld2; p.stats_out = stats_out; p.M = M; p.eps = eps; p.eps_next = eps_next;
    p.partial = nullptr; p.chunks_per_split = 0; p.ntiles = 0;
    hipStream_t st = (hipStream_t)hip_stream;
    if (dtype == DT_F16) return p.wqkv ? launch_mlp_small<f16, true>(p, st) : launch_mlp_small<f16, false>(p, st);
    return p.wqkv ? launch_mlp_small<bf16, true>(p, st) : launch_mlp_small<bf16, false>(p, st);
}

// ---- decoder FFN, first half: partial[s] (M, C) f32 = ReLU(x W1[hs]^T + b1[hs]) W2[:, hs]^T over the hidden slice hs of split s.
// w2_chunked as for lwdetr_mlp_fused (hidden x C chunk-major with the k-slot permutation), w1 (hid, C) plain. C in {256, 384}.
// splits: the number of slabs the kernel writes (query it with lwdetr_ffn_splits, size `partial` as splits * M * C floats).
template <typename T>
static int ffn_dispatch(const MlpParams& p, int C, int hid, hipStream_t st, int* splits_out) {
    switch (C) {
        case 256: return launch_ffn<T, 256, 2>(p, hid, st, splits_out);
        case 384: return launch_ffn<T, 384, 1>(p, hid, st, splits_out);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}
static int ffn_entry(const void* x, long ldx, const void* w1, const float* b1, const void* w2_chunked, float* partial, long M,
                     int C, int hid, int dtype, void* hip_stream, int* splits_out) {
    if (M < 0 || hid <= 0 || hid % 64 != 0 || ldx % 8 != 0) return LWDETR_ERR_BAD_ARG;
    MlpParams p = {};
    p.x = (void*)x; p.ldx = ldx; p.w1 = w1; p.b1 = b1; p.w2p = w2_chunked; p.M = M; p.partial = partial;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F16: return ffn_dispatch<f16>(p, C, hid, st, splits_out);
        case DT_BF16: return ffn_dispatch<bf16>(p, C, hid, st, splits_out);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}
extern "C" int lwdetr_ffn_splits(long M, int C, int hid, int dtype) {
    int s = 0;
    const int rc = ffn_entry(nullptr, 0, nullptr, nullptr, nullptr, nullptr, M, C, hid, dtype, nullptr, &s);
    return rc == LWDETR_OK ? s : -rc;
}
extern "C" int lwdetr_ffn_partial(const void* x, long ldx, const void* w1, const float* b1, const void* w2_chunked, float* partial,
                                  long M, int C, int hid, int dtype, void* hip_stream) {
    if (!x || !w1 || !b1 || !w2_chunked || !partial) return LWDETR_ERR_BAD_ARG;
    if (M == 0) return LWDETR_OK;
    return ffn_entry(x, ldx, w1, b1, w2_chunked, partial, M, C, hid, dtype, hip_stream, nullptr);
}
