// Fused softmax(Q K^T) V for gfx950 - window attention, global attention and the decoder's self-attention.
//
// Replaces the reference's materialised-score formulation (models/backbone/vit.py:130-137: a (B',12,N,N) tensor per
// block; models/attention.py:595-606) with a flash-style kernel: scores never leave registers.
//
// Layout contract (written by the QKV GEMM epilogue, gemm.hip): Q and K are (B, heads, Tp, hd) with Q pre-scaled by
// hd^-0.5 * log2(e); V is stored TRANSPOSED, (B, heads, hd, Tp), because P V contracts over keys and an MFMA operand
// needs its k-run contiguous per lane - the producing GEMM's accumulator layout makes that transposed store free.
// Tokens stay in the ViT's window-major order, so a window is a contiguous range of Twp rows and a global block is
// the whole image: the same kernel serves both (sequence geometry comes from the descriptor).
//
// Per wave: 16*QT queries. The score tile is computed TRANSPOSED, S^T = K Q^T (A operand = K rows, B operand = Q rows):
// the accumulator layout then gives every lane 4 keys x 1 query, which (a) makes the softmax row reduction two
// cross-lane steps (xor 16, 32) and (b) is already the B-operand layout of the second MFMA, O^T = V^T P^T - P goes from
// accumulator registers to MFMA operand with a type conversion only (no LDS, no shuffles). A 32-key step pairs the two
// 16-key tiles into one K=32 MFMA per 16 output channels; the k-slot -> key permutation this implies is applied to the
// V^T operand by loading two 8-byte runs per lane. Waves are independent (no LDS, no barriers): K / V^T tiles are
// re-read through L1/L2, which holds a whole (image, head) slice (1600 x 16 x 2 B x 2 = 100 KB at 640x640).
#include "common.h"
#include <cstdlib>

namespace {

// cross-row reductions on the VALU (v_permlane16_swap / v_permlane32_swap) instead of ds_bpermute through the LDS
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// This file is compiled with -fno-honor-nans (Makefile): otherwise hipcc puts a v_max_f32 x, x canonicalisation in front of
// every fmaxf operand that comes out of an MFMA (it cannot prove the value is not a signalling NaN) - ~15 instructions per
// 8 scores instead of 4 v_max3. Plain VALU does not overlap the matrix pipe on this chip (tools/ubench/issue.hip), so
// softmax instruction count is kernel time. Infinities stay honoured (the key mask is -inf).
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// Tuning builds (tools/attn_ablate.py, -DLWDETR_ATTN_ABL=n, wrong results): 1 = no exp2, 2 = no row max / rescale check,
// 4 = no score MFMAs, 8 = no P V / row-sum MFMAs, 16 = no float -> T conversion of P. Bits combine.
#ifndef LWDETR_ATTN_ABL
#define LWDETR_ATTN_ABL 0
#endif

template <typename T, int HD, int QT, bool SHORT = false>
__global__ __launch_bounds__(256) void attn_kernel(const lwdetr_attn_desc p) {
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    constexpr int NC = HD >= 32 ? HD / 32 : 1;     // contraction chunks for Q K^T
    constexpr int DT = HD / 16;                    // 16-channel output tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, seq = blockIdx.z;
    const int q0 = (blockIdx.x * 4 + wave) * 16 * QT;
    if (q0 >= p.keys_per_seq) return;
    const int b = seq / p.seqs_per_img, w = seq - b * p.seqs_per_img;
    const long tok0 = (long)w * p.seq_tok_stride;
    const long bh = (long)b * p.heads + head;
    const T* __restrict__ Qb = (const T*)p.Q + (bh * p.Tp + tok0) * HD;
    const T* __restrict__ Kb = (const T*)p.K + (bh * p.Tp + tok0) * HD;
    const T* __restrict__ Vb = (const T*)p.VT + bh * HD * (long)p.Tp + tok0;
    const int nkeys = p.keys_per_seq;
    const int last = nkeys - 1;

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[q = l15][d-run of group g]
    V8 q8[QT][NC]; V4 q4[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int q = q0 + t * 16 + l15; q = q < last ? q : last;
        if (HD >= 32) {
#pragma unroll
            for (int c = 0; c < NC; ++c) q8[t][c] = *(const V8*)(Qb + (long)q * HD + c * 32 + g * 8);
        } else {
            q4[t] = *(const V4*)(Qb + (long)q * HD + g * 4);
        }
    }

    // Accumulators: O^T tiles, and the softmax denominator as one extra MFMA against an all-ones A operand (every
    // register of lsum[t] then holds the full row sum of query l15: no VALU adds, no cross-lane reduction).
    f32x4 o[QT][DT], lsum[QT];
    // exponent reference of each query (log2 domain), kept negated and replicated: it is the C operand of every score
    // MFMA as it stands (no per-tile broadcast moves), and only changes in the rare rescale path
    f32x4 negm[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        negm[t] = f32x4{0.f, 0.f, 0.f, 0.f}; lsum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    V8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = from_f32<T>(1.f);

    const bool holes = p.sub_len < p.sub_stride;
    constexpr float RESCALE_THR = 8.f;      // lazy rescaling: keep the old reference while no score exceeds it by 2^8

    // K / V^T operand fragments of one 32-key step; loaded one step ahead (software pipelining: the loads of step i+1
    // are in flight while step i runs its MFMAs and exps)
    struct KV { V8 k8[2][NC]; V4 k4[2]; V8 v[DT]; };
    // Key order inside a 32-key step: score tile kt, row 4g + r  <->  key k0 + 8g + 4kt + r. Any row order is as good as
    // another for S^T = K Q^T, and with this one the 8 k-slots a lane owns in O^T = V^T P^T (slots 0-3 from tile 0, 4-7 from
    // tile 1, rows 4g..4g+3 of each) are the 8 CONSECUTIVE keys k0 + 8g .. + 7: one 16-byte V^T load per lane and step.
    auto load_k = [&](int k0, KV& f) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            int key = k0 + 8 * (l15 >> 2) + 4 * kt + (l15 & 3); key = key < last ? key : last;
            if (HD >= 32) {
#pragma unroll
                for (int c = 0; c < NC; ++c) f.k8[kt][c] = *(const V8*)(Kb + (long)key * HD + c * 32 + g * 8);
            } else {
                f.k4[kt] = *(const V4*)(Kb + (long)key * HD + g * 4);
            }
        }
    };
    auto load_v = [&](int k0, KV& f) {
        int kv = k0 + g * 8;
        kv = kv + 7 < nkeys ? kv : (nkeys - 8);                      // whole run in range or clamped (masked below)
        // clamped only in a ragged last step; keys_per_seq is a multiple of 4, so the run then starts 4 (mod 8) keys early
        // (or >= 8 early: every slot is a key past the end and P = 0 there)
        const bool shift4 = k0 + g * 8 - kv == 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            V8 v = *(const V8*)(Vb + (long)(dt * 16 + l15) * p.Tp + kv);
            if (shift4) {                                            // keep slot e <-> key k0 + 8g + e for the 4 valid keys
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[4 + e];
            }
            f.v[dt] = v;
        }
    };
    auto load_kv = [&](int k0, KV& f) { load_k(k0, f); load_v(k0, f); };

    // score tiles of one query tile against the two key tiles of a step (C operand = -reference)
    auto qk_tile = [&](const KV& f, int t, f32x4 (&st)[2]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x4 a = negm[t];
            if (LWDETR_ATTN_ABL & 4) { a[0] += to_f32<T>(f.k4[kt][0]); st[kt] = a; continue; }
            if (HD >= 32) {
#pragma unroll
                for (int c = 0; c < NC; ++c) a = Mma<T>::k32(f.k8[kt][c], q8[t][c], a);
            } else {
                a = Mma<T>::k16(f.k4[kt], q4[t], a);
            }
            st[kt] = a;
        }
    };
    // mask keys beyond the sequence / pad rows inside it, row maxima, lazy rescale (the reference only moves when some
    // score exceeds it by more than RESCALE_THR - always on the first step; O and the denominator follow exactly once)
    auto softmax_head = [&](f32x4 (&s)[QT][2], int k0) {
        // plain selects: a nested-branch formulation of this block was miscompiled by hipcc 7.2 (inner select dropped)
        if (k0 + 32 > nkeys || holes) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + g * 8 + kt * 4 + r;
                    const int sub = holes ? key % p.sub_stride : 0;
                    const bool ok = (key < nkeys) & (sub < p.sub_len);
#pragma unroll
                    for (int t = 0; t < QT; ++t) s[t][kt][r] = ok ? s[t][kt][r] : -INFINITY;
                }
        }
        float lmax[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            lmax[t] = max3(s[t][0][0], s[t][0][1], s[t][0][2]);
            lmax[t] = max3(lmax[t], s[t][0][3], s[t][1][0]);
            lmax[t] = max3(lmax[t], s[t][1][1], s[t][1][2]);
            lmax[t] = max3(lmax[t], s[t][1][3], s[t][1][3]);
        }
        if (LWDETR_ATTN_ABL & 2) { for (int t = 0; t < QT; ++t) lmax[t] = s[t][0][0]; }
        float lall = lmax[0];
#pragma unroll
        for (int t = 1; t + 1 < QT; t += 2) lall = max3(lall, lmax[t], lmax[t + 1]);
        if (QT % 2 == 0) lall = max3(lall, lmax[QT - 1], lmax[QT - 1]);
        const bool need = k0 == 0 || lall > RESCALE_THR;
        if (__any(need)) {
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                float mx = xor32_max(xor16_max(lmax[t]));           // row max relative to the current reference
                mx = k0 == 0 ? mx : fmaxf(mx, 0.f);                 // the reference never decreases after step 0
                const float alpha = __builtin_amdgcn_exp2f(-mx);
                negm[t] -= mx;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) s[t][kt] -= mx;
                lsum[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[t][dt] *= alpha;
            }
        }
    };
    // P = 2^s of one query tile (already the B operand), row sum and O^T accumulation
    auto pv_tile = [&](const KV& f, int t, const f32x4 (&st)[2]) {
        V8 pf;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (LWDETR_ATTN_ABL & 1) ? st[kt][r] : __builtin_amdgcn_exp2f(st[kt][r]);
                pf[kt * 4 + r] = from_f32<T>(e);
            }
        if (LWDETR_ATTN_ABL & 8) { lsum[t][0] += to_f32<T>(pf[0]) + to_f32<T>(pf[5]); return; }
        lsum[t] = Mma<T>::k32(ones, pf, lsum[t]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[t][dt] = Mma<T>::k32(f.v[dt], pf, o[t][dt]);
    };
    auto step = [&](const KV& f, int k0) {
        f32x4 s[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) qk_tile(f, t, s[t]);
        softmax_head(s, k0);
#pragma unroll
        for (int t = 0; t < QT; ++t) pv_tile(f, t, s[t]);
    };
    // (Tried and measured slower, 185 -> 192 us on the 1600-key shape: issuing the score MFMAs of step i + 1 tile by tile
    // between the exponentials of step i. Transcendentals, MFMAs and plain VALU of the waves of a SIMD add up on this chip
    // whatever the interleaving; only the instruction count moves the time.)

    // The loads of step i + 1 are issued UNCONDITIONALLY (addresses are clamped, a step past the end fetches rows that are
    // never used): with the issue under an `if`, hipcc merges the "nothing new in flight" path into the waits and drains
    // the queue (s_waitcnt vmcnt(0)) right before the MFMAs that need step i's operands - i.e. it waits for the loads it
    // has just issued, one exposed L2 round trip per step.
    // (Short sequences - a 100-key window is 4 steps - keep the conditional form: one wasted fetch per wave costs more
    // there than the drained queue.)
    if constexpr (SHORT) {
        // at most 128 keys (window attention at 640 x 640): all four steps' operands are requested before the first one is used -
        // one memory round trip per wave instead of one per step (the conditional double buffer below drains the queue in front of
        // every step: 28.9 us per launch at config 2 for 91 MB, 3.1 TB/s)
        KV f4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) load_kv(32 * i, f4[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (32 * i >= nkeys) break;
            step(f4[i], 32 * i);
        }
    } else {
    KV fa, fb;
    if (nkeys >= 512) {
        load_kv(0, fa);
        for (int k0 = 0;;) {
            load_kv(k0 + 32, fb);
            step(fa, k0);
            k0 += 32;
            if (k0 >= nkeys) break;
            load_kv(k0 + 32, fa);
            step(fb, k0);
            k0 += 32;
            if (k0 >= nkeys) break;
        }
    } else {
        load_kv(0, fa);
        for (int k0 = 0;;) {
            if (k0 + 32 < nkeys) load_kv(k0 + 32, fb);
            step(fa, k0);
            k0 += 32;
            if (k0 >= nkeys) break;
            if (k0 + 32 < nkeys) load_kv(k0 + 32, fa);
            step(fb, k0);
            k0 += 32;
            if (k0 >= nkeys) break;
        }
    }
    }

    // ---- normalise and store: lane holds channels dt*16 + 4g .. +3 of query q0 + t*16 + l15
    T* __restrict__ out = (T*)p.out;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float l = lsum[t][0];
        const float inv = 1.f / l;
        const int q = q0 + t * 16 + l15;
        if (q < nkeys) {
            T* orow = out + ((long)b * p.Tp + tok0 + q) * p.ldo + head * HD;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                V4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = from_f32<T>(o[t][dt][r] * inv);
                *(V4*)(orow + dt * 16 + g * 4) = ov;
            }
        }
    }
}


// =====================================================================================================================
// Long-sequence variant (global attention of the ViT: 1600 / 3600 keys, 16-bit types). Same math as attn_kernel, built
// around the 32x32x16 MFMA and an LDS ring that the waves of a workgroup share:
//   * a workgroup = NW waves x 32*QT queries of ONE (image, head); K and V^T stream through an NST-deep ring of 64-key
//     stages, filled by global_load_lds DMA in whole 128-byte lines (1 KB pieces), counted s_waitcnt vmcnt + one raw
//     s_barrier per stage. K / V^T enter a CU once per workgroup instead of once per wave, and workgroups of one
//     (image, head) run back to back on one XCD (bijective blockIdx remap) so its L2 serves the re-reads.
//   * scores: S^T = K Q^T as ONE 32x32x16 MFMA per 16 channels (hd = 16: a single instruction per 32 keys x 32 queries,
//     where the 16x16x16 form ran the matrix pipe at half rate). The K rows of a 32-key group enter the A operand with
//     key bits 2 and 3 swapped, which makes the 16 scores a lane owns (query = lane & 31) two runs of 8 CONSECUTIVE keys:
//     registers 8s..8s+7 <-> keys 16s + 8*(lane >> 5) + 0..7. That is exactly the B-operand layout of O^T = V^T P^T with
//     the 32x32x16 MFMA (k-slots 8*(lane >> 5) + e of 16-key slice s): P goes from accumulator to operand with a
//     conversion only - no LDS, no cross-lane moves - and the V^T operand is one 16-byte LDS read per lane.
//   * hd = 16 fills only 16 of the 32 output rows of the P V MFMA: row 16 of the V^T image is all ones (rows 17-31 zero,
//     written once, never touched by the DMA), so the softmax denominator drops out of the same instruction.
//     hd >= 32: denominators by v_dot2c on the packed P (the values the MFMA consumes).
//   * LDS images are row-major with an XOR swizzle of the 16-byte slots (slot ^= (row / rows_per_256B) & (slots - 1)):
//     conflict-free for the 16-lane service groups of ds_read_b128; a DMA piece is lane-linear in LDS, so the swizzle is
//     applied to the SOURCE address (cdna guide rule 21).
// The DMA is issued through inline assembly: hipcc does not count it, so its own waits (for the Q loads, consumed once
// before the loop) never drain the ring; the ring is ordered by the hand-placed counted waits only.
template <typename T> struct Mma32;
template <> struct Mma32<f16> {
    static __device__ __forceinline__ f32x16 k16(f16x8 a, f16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot2(f16x8 p, int i, float c) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 a = {p[2 * i], p[2 * i + 1]}, one = {(f16)1.f, (f16)1.f};
        return __builtin_amdgcn_fdot2(a, one, c, false);
    }
};
template <> struct Mma32<bf16> {
    static __device__ __forceinline__ f32x16 k16(bf16x8 a, bf16x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot2(bf16x8 p, int i, float c) {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        const b2 a = {p[2 * i], p[2 * i + 1]}, one = {(bf16)1.f, (bf16)1.f};
        return __builtin_amdgcn_fdot2_f32_bf16(a, one, c, false);
    }
};

__device__ __attribute__((aligned(16))) unsigned int g_attn_zero16[4];     // source of dummy DMA pieces

// one 1 KB DMA piece: 64 lanes x 16 bytes, LDS destination = wave-uniform base + 16 * lane (m0 written in the statement)
__device__ __forceinline__ void attn_dma16(const void* src, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_base), "v"(src) : "memory");
}
template <int N> __device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int HD, int QT, int NW, int U>
__global__ __launch_bounds__(NW * 64) void attn_lds_kernel(const lwdetr_attn_desc p) {
    typedef typename Vec<T>::v8 V8;
    constexpr int NC = HD / 16;                        // 16-deep contraction chunks of K Q^T
    constexpr int RT = HD > 32 ? HD / 32 : 1;          // 32-row tiles of O^T
    constexpr int RB = HD * 2;                         // bytes of a K row
    constexpr int CPR = RB / 16, RPB = 256 / RB;       // 16-byte slots per K row, K rows per 256 bytes
    constexpr int KB = 64;                             // keys per stage (one 128-byte line of every V^T row)
    constexpr int KIMG = KB * RB, VIMG = RT * 32 * 128, STAGE = KIMG + VIMG;
    constexpr int NKP = KIMG / 1024, NVP = HD * 128 / 1024, NP = NKP + NVP;   // DMA pieces per 64-key block (real V^T rows only)
    // A ring step (one wait + one barrier + one refill) moves U 64-key blocks: the per-step costs - every wave's DMA issue (dummies
    // included), the counted wait, the barrier - are paid once per 2 U score tiles of a wave instead of once per 2
    constexpr int SSTAGE = U * STAGE;                  // bytes of a ring step
    constexpr int PPW = (U * NP + NW - 1) / NW;        // pieces per wave and step (the same for every wave: dummies fill up)
    constexpr int NST = 3;
    constexpr int QW = 32 * QT, QWG = QW * NW;
    constexpr float RESCALE_THR = 8.f;
    __shared__ __attribute__((aligned(16))) char smem[NST * SSTAGE + 1024];    // ring + landing area of dummy pieces

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int nkeys = p.keys_per_seq;
    const int wgs_per_seq = (nkeys + QWG - 1) / QWG;
    int wid;
    {   // workgroups of one (image, head) - and of one image - run on the same XCD, next to each other in time
        const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int qblk = wid % wgs_per_seq;
    const int rest = wid / wgs_per_seq;
    const int head = rest % p.heads, seq = rest / p.heads;
    const int b = seq / p.seqs_per_img, w = seq - b * p.seqs_per_img;
    const long tok0 = (long)w * p.seq_tok_stride;
    const long bh = (long)b * p.heads + head;
    const T* __restrict__ Qb = (const T*)p.Q + (bh * p.Tp + tok0) * HD;
    const T* __restrict__ Kb = (const T*)p.K + (bh * p.Tp + tok0) * HD;
    const T* __restrict__ Vb = (const T*)p.VT + bh * HD * (long)p.Tp + tok0;
    const int q0 = qblk * QWG + wave * QW;
    const bool active = q0 < nkeys;                    // wave-uniform; idle waves still feed the ring and hit the barriers
    const int nblk = (nkeys + KB - 1) / KB;            // 64-key blocks; a ragged last one is masked in the softmax (keys >= nkeys)
    const int nstep = (nblk + U - 1) / U;              // ring steps (blocks past the sequence: dummy pieces, no compute)

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q = m][16c + 8h .. + 7]
    V8 qf[QT][NC];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int q = q0 + t * 32 + m; q = q < nkeys ? q : nkeys - 1;
#pragma unroll
        for (int c = 0; c < NC; ++c) qf[t][c] = *(const V8*)(Qb + (long)q * HD + c * 16 + h * 8);
    }

    // ---- DMA piece table of this wave: piece i = wave + j * NW of a stage (K pieces first, then V^T, then dummies)
    // Stages are whole 64-key blocks; in the LAST block of a ragged sequence the rows / 8-key runs past the sequence are
    // not fetched through the block formula (they may lie outside the tensors): K lanes clamp to the last key, V^T lanes
    // whose run starts at or past nkeys read the zero page. A run that straddles nkeys (nkeys % 8 == 4) is fetched whole -
    // 8 bytes past the sequence, inside the V^T tensor except for its very last row: the caller guarantees that slack
    // (lwdetr_attn_desc.vt_slack), otherwise such shapes stay on attn_kernel.
    const T* src0[PPW]; const T* srct[PPW]; int sstep[PPW]; unsigned dst0[PPW]; bool real[PPW]; int sub[PPW];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
    const unsigned dummy_dst = lds0 + NST * SSTAGE;
    const int kt0 = (nblk - 1) * KB;                    // first key of the last block
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int ii = wave + j * NW;
        const int i = ii % NP;                          // piece of its 64-key block
        sub[j] = ii / NP;                               // block of the step
        if (ii >= U * NP) {
            src0[j] = (const T*)g_attn_zero16; srct[j] = src0[j]; sstep[j] = 0; dst0[j] = 0; real[j] = false;
        } else if (i < NKP) {
            const int off = i * 1024 + lane * 16;                       // byte offset inside the K image
            const int key = off / RB, slot = (off % RB) / 16;
            const int chunk = slot ^ ((key / RPB) & (CPR - 1));
            src0[j] = Kb + key * HD + chunk * 8; sstep[j] = KB * HD; dst0[j] = sub[j] * STAGE + i * 1024; real[j] = true;
            const int kc = kt0 + key < nkeys ? kt0 + key : nkeys - 1;
            srct[j] = Kb + (long)kc * HD + chunk * 8;
        } else {
            const int v = i - NKP;
            const int row = 8 * v + (lane >> 3), slot = lane & 7;
            const int chunk = slot ^ ((row >> 1) & 7);
            src0[j] = Vb + (long)row * p.Tp + chunk * 8; sstep[j] = KB; dst0[j] = sub[j] * STAGE + KIMG + v * 1024; real[j] = true;
            srct[j] = kt0 + chunk * 8 < nkeys ? src0[j] + kt0 : (const T*)g_attn_zero16;
        }
    }
    auto issue = [&](int step, int slot_) {            // always PPW pieces per wave: the wait counts are constants
        const unsigned sb = lds0 + slot_ * SSTAGE;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int blk = step * U + sub[j];
            const bool live = real[j] && blk < nblk;    // wave-uniform
            const T* src = !live ? (const T*)g_attn_zero16 : (blk == nblk - 1 ? srct[j] : src0[j] + (long)blk * sstep[j]);
            attn_dma16(src, __builtin_amdgcn_readfirstlane(live ? sb + dst0[j] : dummy_dst));
        }
    };

    // ---- constant rows of the V^T images (hd = 16): row 16 = ones (the denominator), rows 17..31 = zero
    if (HD == 16) {
        V8 one8, zero8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { one8[e] = from_f32<T>(1.f); zero8[e] = from_f32<T>(0.f); }
        for (int i = threadIdx.x; i < NST * U * 16 * 8; i += NW * 64) {   // 16 rows x 8 slots per 64-key image
            const int st = i / 128, r = (i % 128) / 8, sl = i % 8;
            *(V8*)(smem + st * STAGE + KIMG + (16 + r) * 128 + sl * 16) = r == 0 ? one8 : zero8;
        }
    }

    // ---- per-lane LDS read offsets. K: key pi(m) = m with bits 2 and 3 swapped; V^T: row = m (+ 32 rt)
    const int km = (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1);
    int kofs[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) kofs[c] = km * RB + (((2 * c + h) ^ ((km / RPB) & (CPR - 1))) << 4);
    int vofs[2][2];                                    // [32-key group][16-key slice]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) vofs[t][s] = KIMG + m * 128 + (((4 * t + 2 * s + h) ^ ((m >> 1) & 7)) << 4);

    // ---- accumulators. negm = -reference of the lane's query, replicated: the C operand of the score MFMA
    f32x16 o[QT][RT], negm[QT];
    float lsum[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        lsum[t] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) negm[t][e] = 0.f;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[t][r][e] = 0.f;
    }
    const bool holes = p.sub_len < p.sub_stride;
    int res0 = 0;                                      // (first key of the current 32-key group) mod sub_stride

    // ---- ring prologue, then make sure Q is in registers before the loop (see the header: the compiler's wait for these
    // loads must not end up inside the loop, where it would drain the ring every iteration)
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s, s);
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int c = 0; c < NC; ++c) asm volatile("" :: "v"(qf[t][c]));

    int slot = 0;
    for (int ks = 0; ks < nstep; ++ks) {
        // every K / V^T fragment read of stage kb - 1 has returned before this wave lets the others refill that buffer: hipcc does
        // not see the DMA and may sink the last MFMAs of a stage (and the lgkmcnt wait of their operands) below the barrier
        // (the 3x3 convolution kernel was caught by exactly this beside another stream's LDS-heavy kernels, gemm.hip)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        attn_wait_vmcnt<(NST - 2) * PPW>();            // step ks has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                  // ... and everybody's; everybody has also left step ks - 1
        {
            int ns = slot + NST - 1; ns = ns >= NST ? ns - NST : ns;
            issue(ks + NST - 1, ns);                   // refill the buffer step ks - 1 used
        }
        const char* sS = smem + slot * SSTAGE;
        slot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
        const int kb = ks * U + u;
        if (kb >= nblk) break;                         // blocks past the sequence (wave-uniform)
        const char* sK = sS + u * STAGE;
        if (active) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                V8 kf[NC], vf[2][RT];
#pragma unroll
                for (int c = 0; c < NC; ++c) kf[c] = *(const V8*)(sK + t * 32 * RB + kofs[c]);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int r = 0; r < RT; ++r) vf[s][r] = *(const V8*)(sK + r * 4096 + vofs[t][s]);
                f32x16 sc[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    f32x16 a = negm[qt];
#pragma unroll
                    for (int c = 0; c < NC; ++c) a = Mma32<T>::k16(kf[c], qf[qt][c], a);
                    sc[qt] = a;
                }
                // pad rows inside the sequence (960 x 960: 225 tokens in 228-row windows): register e <-> key
                // k0 + 16 (e >> 3) + 8 h + (e & 7)
                const int k0 = kb * KB + t * 32;
                if ((holes && res0 + 31 >= p.sub_len) || k0 + 32 > nkeys) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ke = 16 * (e >> 3) + 8 * h + (e & 7);
                        int rr = res0 + ke;
                        rr = rr >= p.sub_stride ? rr - p.sub_stride : rr;
                        const bool ok = rr < p.sub_len && k0 + ke < nkeys;
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) sc[qt][e] = ok ? sc[qt][e] : -INFINITY;
                    }
                }
                res0 += 32; res0 = res0 >= p.sub_stride ? res0 - p.sub_stride : res0;
                // lazy rescale: the reference only moves when a score exceeds it by 2^RESCALE_THR (always on the first group)
                float lmax[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    float lm = max3(sc[qt][0], sc[qt][1], sc[qt][2]);
#pragma unroll
                    for (int e = 3; e < 15; e += 2) lm = max3(lm, sc[qt][e], sc[qt][e + 1]);
                    lmax[qt] = __builtin_fmaxf(lm, sc[qt][15]);
                }
                float lall = lmax[0];
#pragma unroll
                for (int qt = 1; qt < QT; ++qt) lall = __builtin_fmaxf(lall, lmax[qt]);
                const bool first = kb == 0 && t == 0;
                if (__any(first || lall > RESCALE_THR)) {
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        float mx = xor32_max(lmax[qt]);                 // both halves of the query's 32 keys
                        mx = first ? mx : fmaxf(mx, 0.f);               // the reference never decreases
                        const float alpha = __builtin_amdgcn_exp2f(-mx);
#pragma unroll
                        for (int e = 0; e < 16; ++e) { negm[qt][e] -= mx; sc[qt][e] -= mx; }
                        lsum[qt] *= alpha;
#pragma unroll
                        for (int r = 0; r < RT; ++r)
#pragma unroll
                            for (int e = 0; e < 16; ++e) o[qt][r][e] *= alpha;
                    }
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    V8 pf[2];
#pragma unroll
                    for (int e = 0; e < 16; ++e) pf[e >> 3][e & 7] = from_f32<T>(__builtin_amdgcn_exp2f(sc[qt][e]));
                    if (HD > 16) {
#pragma unroll
                        for (int s = 0; s < 2; ++s)
#pragma unroll
                            for (int i = 0; i < 4; ++i) lsum[qt] = Mma32<T>::dot2(pf[s], i, lsum[qt]);
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int r = 0; r < RT; ++r) o[qt][r] = Mma32<T>::k16(vf[s][r], pf[s], o[qt][r]);
                }
            }
        } else {
            res0 += 64; res0 %= p.sub_stride;
        }
        }
    }
    attn_wait_vmcnt<0>();                              // the dummy tail pieces
    if (!active) return;

    // ---- normalise and store. O^T register e of tile r <-> channel 32 r + 8 (e >> 2) + 4 h + (e & 3) of query m; a
    // permlane32 swap pairs the halves so that every lane stores 16 contiguous bytes (channels 32 r + 16 j + 8 h .. + 7)
    T* __restrict__ out = (T*)p.out;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l;
        if (HD == 16) {
            const unsigned x = __float_as_uint(o[qt][0][8]);            // row 16 (lanes h = 0): sum of P
            const auto rr = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            l = __uint_as_float(rr[0]);
        } else {
            l = xor32_sum(lsum[qt]);
        }
        const float inv = 1.f / l;
        const int q = q0 + qt * 32 + m;
        T* orow = out + ((long)b * p.Tp + tok0 + q) * p.ldo + head * HD;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int j = 0; j < (HD == 16 ? 1 : 2); ++j) {
                typedef typename Vec<T>::v4 V4;
                V4 a4, b4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a4[e] = from_f32<T>(o[qt][r][8 * j + e] * inv);
                    b4[e] = from_f32<T>(o[qt][r][8 * j + 4 + e] * inv);
                }
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                u2 au = __builtin_bit_cast(u2, a4), bu = __builtin_bit_cast(u2, b4);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(au[d], bu[d], false, false);
                    au[d] = sw[0]; bu[d] = sw[1];
                }
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 st = {au[0], au[1], bu[0], bu[1]};
                if (q < nkeys) *(u4*)(orow + r * 32 + j * 16 + h * 8) = st;
            }
    }
}

template <typename T, int HD, int QT, int NW, int U = 1>
int launch_lds(const lwdetr_attn_desc& p, hipStream_t st) {
    if constexpr (3 * U * (64 * HD * 2 + (HD > 32 ? HD / 32 : 1) * 32 * 128) + 1024 > 65536) return LWDETR_ERR_UNSUPPORTED;   // two workgroups per CU
    else {
    const int wgs_per_seq = (p.keys_per_seq + 32 * QT * NW - 1) / (32 * QT * NW);
    const long nwg = (long)wgs_per_seq * p.heads * p.B * p.seqs_per_img;
    const double nseq = (double)p.B * p.seqs_per_img;
    const double flops = 4.0 * nseq * p.heads * (double)p.keys_per_seq * p.keys_per_seq * HD;
    const double bytes = 4.0 * nseq * p.heads * p.keys_per_seq * HD * sizeof(T);
    ProfScope ps(p.kind == 0 ? KID_ATTN_WINDOW : (p.kind == 1 ? KID_ATTN_GLOBAL : KID_ATTN_DECODER), flops, bytes, st);
    hipLaunchKernelGGL((attn_lds_kernel<T, HD, QT, NW, U>), dim3((unsigned)nwg), dim3(NW * 64), 0, st, p);
    return lwdetr_check_launch();
    }
}

// The LDS-ring kernel serves long, 64-aligned sequences of the 16-bit types; everything else (windows, the decoder's 300
// queries, f32 parity mode) stays on attn_kernel. LWDETR_ATTN_LDS=0 forces attn_kernel, LWDETR_ATTN_LDS_CFG=<100 QT + NW>
// (108, 110, 204, 205, 208) picks a tuning variant.
// Measured on MI355X (tools/attn_bench.py, us per launch, attn_kernel | 2x5 | 2x4 | 1x8 | 2x8):
//   small  B32 f16  hd16 1600 keys:  191 | 238 | 201 | 181 | 199        medium B64 bf16 hd32: 475 | 599 | 493 | 446 | 506
//   large  B32 f16  hd32 1600 keys:  261 | 312 | 269 | 235 | 252        xlarge B16 f16 hd64 3648 keys: 2226 | 1068 | 895 | 848 | 827
// hd <= 32 is bound by the exponentials (v_exp_f32 issues at 8.5 cycles per wave and does NOT overlap other VALU work on
// this chip, tools/ubench/overlap.hip): 32 queries per wave keep the register count at 84-92 (5 waves per SIMD), which is
// what hides the LDS / barrier latencies there; hd = 64 is matrix-bound and prefers the K / V^T reuse of 64 queries per wave.
int g_attn_lds_cfg = 0;         // lwdetr_attention_tuning_cfg(): overrides the environment / default (tests)
}  // namespace
// The hd 16 instantiations of the ring kernel live in a second object built from THIS file with -DLWDETR_ATTN_HD16_TU -fno-slp-vectorize
// (Makefile: attention_hd16.o): hipcc's SLP pass packs the softmax's f32 adds / multiplies into v_pk_*_f32, one of which costs what five plain
// instructions cost beside an MFMA (tools/microbench/filler_bench.hip) - global attention at hd 16 is VALU-bound and loses 2 % to them
// (profiles/r6f_*: 79.2 -> 77.6 us, config 2 +0.7 %), hd 64 is matrix-bound and WINS 0.9 % with the packed forms (xlarge), hd 32 does not care.
extern "C" __attribute__((visibility("hidden"))) int lwdetr_attn_lds_hd16_impl(const lwdetr_attn_desc* p, int dtype, void* hip_stream, int cfg);
namespace {
template <typename T, int HD>
int launch_lds_cfg(const lwdetr_attn_desc& p, hipStream_t st) {
    int c = g_attn_lds_cfg ? g_attn_lds_cfg : (int)lwdetr_knob(KNOB_ATTN_LDS_CFG, 0);
#ifndef LWDETR_ATTN_HD16_TU
    if constexpr (HD == 16) return lwdetr_attn_lds_hd16_impl(&p, std::is_same<T, f16>::value ? DT_F16 : DT_BF16, (void*)st, c);
    else {
#else
    {
#endif
    // a 100-key window = 4 waves of 32 queries. hd 16 (round 4): 128-query workgroups (13 per 1600-query image and head, the last
    // half idle; 4-5 resident per CU) beat 256-query ones (7, the last three quarters idle; 2 per CU): 161 vs 180 us on the bench
    // shape, +1.6 % on config 2 (profiles/r4f_attn_global_workgroup_sizes.txt); hd 32 is indifferent, hd 64 wants the K / V^T reuse
    if (!c) c = p.keys_per_seq <= 128 || HD == 16 ? 104 : (HD >= 64 && p.keys_per_seq >= 512 ? 208 : 108);
    switch (c) {
        case 102: return launch_lds<T, HD, 1, 2>(p, st);
        case 103: return launch_lds<T, HD, 1, 3>(p, st);
        case 104: return launch_lds<T, HD, 1, 4>(p, st);
        case 105: return launch_lds<T, HD, 1, 5>(p, st);
        case 106: return launch_lds<T, HD, 1, 6>(p, st);
        case 108: return launch_lds<T, HD, 1, 8>(p, st);
        case 110: return launch_lds<T, HD, 1, 10>(p, st);
        case 204: return launch_lds<T, HD, 2, 4>(p, st);
        case 205: return launch_lds<T, HD, 2, 5>(p, st);
        case 208: return launch_lds<T, HD, 2, 8>(p, st);
        case 1104: return launch_lds<T, HD, 1, 4, 2>(p, st);
        case 1105: return launch_lds<T, HD, 1, 5, 2>(p, st);
        case 1108: return launch_lds<T, HD, 1, 8, 2>(p, st);       // 1000 (U - 1) + 100 QT + NW: 128 keys per ring step
        case 1110: return launch_lds<T, HD, 1, 10, 2>(p, st);
        case 3108: return launch_lds<T, HD, 1, 8, 4>(p, st);       // 256 keys per ring step
        case 1208: return launch_lds<T, HD, 2, 8, 2>(p, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
    }
}
template <typename T, int HD> struct LdsPath {
    static bool ok(const lwdetr_attn_desc&) { return false; }
    static int go(const lwdetr_attn_desc&, hipStream_t) { return LWDETR_ERR_UNSUPPORTED; }
};
template <int HD> struct LdsPath<f16, HD> {
    static bool ok(const lwdetr_attn_desc& p) { return true; }
    static int go(const lwdetr_attn_desc& p, hipStream_t st) { return launch_lds_cfg<f16, HD>(p, st); }
};
template <int HD> struct LdsPath<bf16, HD> {
    static bool ok(const lwdetr_attn_desc& p) { return true; }
    static int go(const lwdetr_attn_desc& p, hipStream_t st) { return launch_lds_cfg<bf16, HD>(p, st); }
};
// Sequences the LDS-ring kernel takes: >= 64 keys (a ViT window or a whole image), 16-byte aligned K rows and outputs
// (always), V^T runs that start on 8-byte boundaries (global_load_lds has dword alignment), masks with sub_stride >= 64.
// A sequence length that is not a multiple of 8 makes the last 16-byte V^T run straddle the sequence end: taken only when
// the caller vouches for 8 readable bytes after the V^T tensor (vt_slack). LWDETR_ATTN_LDS: 0 = never, 1 = long sequences
// only (>= 512 keys), default (2) = also windows of >= 192 keys, 3 = everything >= 64 keys (tests).
int g_attn_lds_mode = -1;        // lwdetr_attention_tuning(): overrides the environment / default (tests)
static bool lds_path_applies(const lwdetr_attn_desc& p) {
    const int mode = g_attn_lds_mode >= 0 ? g_attn_lds_mode : (int)lwdetr_knob(KNOB_ATTN_LDS, 2);
    if (mode == 0) return false;
    // measured (tools/attn_bench.py, us per launch, attn_kernel | LDS ring): 228-key windows hd64 B16: 240 | 131; 100-key windows
    // hd16 B32: 36 | 54, hd32 B64: 113 | 124 - a 100-key sequence is over after two stages, the ring never pays for its start-up
    if (p.keys_per_seq < (mode == 1 ? 512 : (mode == 3 ? 64 : 192))) return false;
    if (p.keys_per_seq % 8 != 0 && !p.vt_slack) return false;
    // a grid that cannot fill the chip (single image: 7 x 12 workgroups of 256 queries) is faster on attn_kernel's 128-query
    // workgroups: 21.7 vs 27.0 us at B = 1, hd 16, 1600 keys (tools/attn_bench.py small_b1_f16)
    // (round 5: at hd 16 the ring kernel runs 128-query workgroups since round 4 - 156 of them for one image - and wins from there:
    // single-image p50 0.867 -> 0.857 ms with it on the four global-attention launches, tools/lat_bs1.py, profiles/r5e_*)
    if (mode != 3 && (p.hd == 16 ? (long)((p.keys_per_seq + 127) / 128) * p.heads * p.B * p.seqs_per_img < 128
                                 : (long)((p.keys_per_seq + 255) / 256) * p.heads * p.B * p.seqs_per_img < 256)) return false;
    return p.keys_per_seq % 4 == 0 && p.seq_tok_stride % 4 == 0 && p.Tp % 4 == 0 && p.ldo % 8 == 0 &&
           (p.sub_len == p.sub_stride || p.sub_stride >= 64);
}

// =====================================================================================================================
// Short-sequence variant (window attention at 640 x 640: 100 keys; any sequence of <= 128 keys, hd 16 / 32, 16-bit types).
// ONE WAVE owns a whole (sequence, head): its K and V^T live in registers as MFMA operand fragments for the whole kernel
// (4 + 8 fragments at hd 16), each 32-query tile is S^T = K Q^T (32x32x16, the contraction IS hd at hd 16: no padding) ->
// row maximum over the lane's registers + one half-wave exchange -> exp2 -> P packed in place as the B operand of
// O^T = V^T P^T (k-slot (h, j) of 16-key chunk c is key 16 c + 8 (j >> 2) + 4 h + (j & 3): V^T is loaded in that order) ->
// normalise -> half-wave exchange to 8 consecutive channels per lane -> 16-byte stores. At hd 16 rows 16-31 of the V^T operand
// are free: row 16 is all ones, so the softmax denominator drops out of the same MFMAs. attn_kernel spreads a 100-key window
// over 4 waves that each re-load K and V^T and run 4 dependent load -> MFMA -> softmax -> MFMA steps (19 us at B = 16 for
// 40 MB of q / k / v / out, 2.4x the HBM floor; tools/attn_bench.py); here a wave issues all its loads at once.
// Ablation builds for tuning (-DLWDETR_AW_ABL=bits, results are WRONG; never in the product): 1 = no output stores, 2 = no V^T loads,
// 4 = no K / Q loads, 8 = no exp2 (profiles/r5c_*).
#ifndef LWDETR_AW_ABL
#define LWDETR_AW_ABL 0
#endif
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_win_kernel(const lwdetr_attn_desc p) {
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    constexpr int NC = HD / 16;                         // 16-deep chunks of the Q K^T contraction
    constexpr int RT = HD / 32 > 0 ? HD / 32 : 1;      // 32-row tiles of the output channels (hd 16: half a tile + the ones row)
    static_assert(sizeof(T) == 2 && (HD == 16 || HD == 32), "16-bit types, hd 16 / 32");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
    const long npairs = (long)p.B * p.heads * p.seqs_per_img;
    const long pair = (long)blockIdx.x * 4 + wave;      // (image, head, window): the four waves of a workgroup are neighbours in memory
    if (pair >= npairs) return;
    const int w = (int)(pair % p.seqs_per_img);
    const long bh = pair / p.seqs_per_img;
    const int head = (int)(bh % p.heads), b = (int)(bh / p.heads);
    const long tok0 = (long)w * p.seq_tok_stride;
    const T* __restrict__ Qb = (const T*)p.Q + (bh * p.Tp + tok0) * HD;
    const T* __restrict__ Kb = (const T*)p.K + (bh * p.Tp + tok0) * HD;
    const T* __restrict__ Vb = (const T*)p.VT + bh * HD * (long)p.Tp + tok0;
    const int nkeys = p.keys_per_seq, last = nkeys - 1;
    const int nvalid = p.sub_len < nkeys ? p.sub_len : nkeys;        // pad rows sit behind the real tokens (one sub-window)
    const int nqt = (nkeys + 31) >> 5, nkt = (nvalid + 31) >> 5, nch = (nvalid + 15) >> 4;

    // ---- everything this wave will ever read, issued at once
    V8 kf[4][NC], vf[8][RT], qf[4][NC];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        int key = kt * 32 + l31; key = key < last ? key : last;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#if LWDETR_AW_ABL & 4
            for (int e = 0; e < 8; ++e) kf[kt][c][e] = from_f32<T>(0.01f * (key + e));
#else
            kf[kt][c] = *(const V8*)(Kb + (long)key * HD + c * 16 + h * 8);
#endif
        }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        int q = qt * 32 + l31; q = q < last ? q : last;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#if LWDETR_AW_ABL & 4
            for (int e = 0; e < 8; ++e) qf[qt][c][e] = from_f32<T>(0.02f * (q - e));
#else
            qf[qt][c] = *(const V8*)(Qb + (long)q * HD + c * 16 + h * 8);
#endif
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int row = r * 32 + l31;              // output channel (hd 16: rows 16-31 are not channels)
            V8 v;
            if (HD == 16 && row >= 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(row == 16 ? 1.f : 0.f);
            } else {
                // runs of 4 keys; a run past the end is clamped to the last run (finite values; its P is exactly 0)
                int k0 = 16 * c + 4 * h, k1 = k0 + 8;
                k0 = k0 + 4 <= nkeys ? k0 : nkeys - 4; k1 = k1 + 4 <= nkeys ? k1 : nkeys - 4;
#if LWDETR_AW_ABL & 2
                for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(0.03f * (row + k0 + e));
#else
                const V4 lo = *(const V4*)(Vb + (long)row * p.Tp + k0), hi = *(const V4*)(Vb + (long)row * p.Tp + k1);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
#endif
            }
            vf[c][r] = v;
        }
    }

    T* __restrict__ out = (T*)p.out;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        if (qt >= nqt) break;                           // wave-uniform
        // ---- S^T tile by tile: register 4 b + e of lane (query l31, h) is key 32 kt + 8 b + 4 h + e
        f32x16 sc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt >= nkt) break;
            f32x16 a;
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) a = Mma32<T>::k16(kf[kt][c], qf[qt][c], a);
            if (kt * 32 + 32 > nvalid) {                // the ragged last tile: keys past the real tokens
                asm volatile("; ragged key tile");      // keeps this a branch: hipcc otherwise runs the 16 compares + selects on every tile
#pragma unroll
                for (int e = 0; e < 16; ++e) a[e] = kt * 32 + 8 * (e >> 2) + 4 * h + (e & 3) < nvalid ? a[e] : -INFINITY;
            }
            sc[kt] = a;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt >= nkt) break;
            float lm = max3(sc[kt][0], sc[kt][1], sc[kt][2]);
#pragma unroll
            for (int e = 3; e < 15; e += 2) lm = max3(lm, sc[kt][e], sc[kt][e + 1]);
            mx = max3(mx, lm, sc[kt][15]);
        }
        mx = xor32_max(mx);                             // both halves of the query's keys
        // ---- P = exp2(S - max), packed per 16-key chunk as the B operand; O^T = V^T P^T
        f32x16 o[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[r][e] = 0.f;
        float lsum = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c >= nch) break;
            V8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#if LWDETR_AW_ABL & 8
                const float pe = sc[c >> 1][8 * (c & 1) + e] - mx;
#else
                const float pe = __builtin_amdgcn_exp2f(sc[c >> 1][8 * (c & 1) + e] - mx);
#endif
                if (HD > 16) lsum += pe;
                pf[e] = from_f32<T>(pe);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) o[r] = Mma32<T>::k16(vf[c][r], pf, o[r]);
        }
        // ---- normalise; register 4 b + e of lane (query, h) is channel 8 b + 4 h + e (hd 16: the ones row is channel 16 = b 2, h 0, e 0)
        float inv;
        if (HD == 16) {
            const auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[0][8]), __float_as_uint(o[0][8]), false, false);
            inv = 1.f / (h == 0 ? o[0][8] : __uint_as_float(rr[0]));       // r[0] of an upper lane is the lower half's value
        } else {
            inv = 1.f / xor32_sum(lsum);
        }
        const int q = qt * 32 + l31;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
#pragma unroll
            for (int bp = 0; bp < (HD == 16 ? 1 : 2); ++bp) {      // pairs of 4-channel blocks (b = 2 bp, 2 bp + 1): channels 16 bp + 4 h + .., 16 bp + 8 + 4 h + ..
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                V4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = from_f32<T>(o[r][8 * bp + e] * inv); hi[e] = from_f32<T>(o[r][8 * bp + 4 + e] * inv); }
                const u32x2 lu = __builtin_bit_cast(u32x2, lo), hu = __builtin_bit_cast(u32x2, hi);
                // lower lanes keep lo (channels 0-3) and take the upper lanes' lo (4-7); upper lanes take the lower lanes' hi (8-11) and keep hi
                const auto s0 = __builtin_amdgcn_permlane32_swap(lu[0], hu[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(lu[1], hu[1], false, false);
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                const u32x4v st = {s0[0], s1[0], s0[1], s1[1]};
#if LWDETR_AW_ABL & 1
                if (q < nkeys && p.B < 0)
#else
                if (q < nkeys)
#endif
                    *(u32x4v*)(out + ((long)b * p.Tp + tok0 + q) * p.ldo + head * HD + r * 32 + 16 * bp + 8 * h) = st;
            }
        }
    }
}

template <typename T, int HD>
int launch_win(const lwdetr_attn_desc& p, hipStream_t st) {
    const long npairs = (long)p.B * p.heads * p.seqs_per_img;
    const double nseq = (double)p.B * p.seqs_per_img;
    const double flops = 4.0 * nseq * p.heads * (double)p.keys_per_seq * p.keys_per_seq * HD;
    const double bytes = 4.0 * nseq * p.heads * p.keys_per_seq * HD * sizeof(T);
    const int kid = p.kind == 0 ? KID_ATTN_WINDOW : (p.kind == 1 ? KID_ATTN_GLOBAL : KID_ATTN_DECODER);
    ProfScope ps(kid, flops, bytes, st);
    hipLaunchKernelGGL((attn_win_kernel<T, HD>), dim3((unsigned)((npairs + 3) / 4)), dim3(256), 0, st, p);
    return lwdetr_check_launch();
}

// =====================================================================================================================
// Window-tile variant (round 5; hd 16, windows of <= 128 keys - window attention of the C = 192 models at 640 x 640): ONE WORKGROUP
// owns a whole (image, window) with all of its heads, and the window's tiles go through LDS:
//   in : V^T of the window, all heads - (heads x 16) rows of `keys` values, 200-byte pieces of rows that lie 2 Tp bytes apart in
//        memory - is staged in LDS by all 256 threads in coalesced 8-byte pieces (attn_win_kernel fetches them as MFMA fragments:
//        14 load instructions per head with half the lanes idle, each touching 16 different lines); the A-operand fragments of
//        O^T = V^T P^T are then conflict-free 8-byte LDS reads (row stride 52 dwords). Q and K of a (window, head) are contiguous
//        3.2 KB runs already and stay direct register loads (whole 1 KB lines per instruction), prefetched one head ahead.
//   out: every wave (heads w, w + 4, w + 8) writes its 16 channels of the window's rows into an LDS tile [keys][C]; after one
//        barrier the workgroup stores the tile with full 16-byte-per-lane rows - the window's output is ONE contiguous run of
//        keys x C x 2 bytes (window-major tokens, ldo = C) - instead of 32-byte pieces 2 C bytes apart per (token, head).
// Arithmetic, masks and rounding points are attn_win_kernel's (the same MFMA sequence per (window, head)); results are bit-identical.
#ifdef LWDETR_EXPERIMENTS      // see gemm.hip: measured-slower forms are not in the default build
template <typename T>
__global__ __launch_bounds__(256) void attn_wtile_kernel(const lwdetr_attn_desc p) {
    constexpr int HD = 16;
    typedef typename Vec<T>::v8 V8;
    typedef typename Vec<T>::v4 V4;
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    static_assert(sizeof(T) == 2, "16-bit types");
    extern __shared__ __attribute__((aligned(16))) char wt_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    const int w = (int)(blockIdx.x % (unsigned)p.seqs_per_img), b = (int)(blockIdx.x / (unsigned)p.seqs_per_img);
    const long tok0 = (long)w * p.seq_tok_stride;
    const int nkeys = p.keys_per_seq, last = nkeys - 1;
    const int nvalid = p.sub_len < nkeys ? p.sub_len : nkeys;        // pad rows sit behind the real tokens (one sub-window)
    const int nqt = (nkeys + 31) >> 5, nkt = (nvalid + 31) >> 5, nch = (nvalid + 15) >> 4;
    const int C = p.heads * HD;
    const int VLD = ((nkeys + 7) & ~7) + 8;                          // V^T row stride in elements: 104 + 8 = 56 dwords at 100 keys
    const int OLD = C + 8;                                           // output tile row stride in elements (16-byte multiple)
    T* vs = (T*)wt_smem;                                             // [heads * 16][VLD]
    T* os = vs + (size_t)C * VLD;                                    // [nkeys][OLD]

    // ---- Q / K of this wave's first head: issued before anything else
    auto load_qk = [&](int head, V8 (&kf)[4], V8 (&qf)[4]) {
        const long bh = (long)b * p.heads + head;
        const T* __restrict__ Qb = (const T*)p.Q + (bh * p.Tp + tok0) * HD;
        const T* __restrict__ Kb = (const T*)p.K + (bh * p.Tp + tok0) * HD;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int r = t * 32 + l31; r = r < last ? r : last;
            kf[t] = *(const V8*)(Kb + (long)r * HD + h * 8);
            qf[t] = *(const V8*)(Qb + (long)r * HD + h * 8);
        }
    };
    V8 kf[4], qf[4];
    int head = wave;
    if (head < p.heads) load_qk(head, kf, qf);
    // ---- stage V^T of the window (all heads): thread -> (row, 8-byte piece), consecutive threads = consecutive pieces of a row
    {
        const int ppr = nkeys >> 2, total = C * ppr;                 // pieces per row (keys are a multiple of 4)
        const T* __restrict__ Vw = (const T*)p.VT + (long)b * C * (long)p.Tp + tok0;
        for (int i = tid; i < total; i += 256) {
            const int row = i / ppr, pc = i - row * ppr;
            *(V4*)(vs + (size_t)row * VLD + 4 * pc) = *(const V4*)(Vw + (long)row * p.Tp + 4 * pc);
        }
    }
    __syncthreads();

    while (head < p.heads) {
        const int nh = head + 4;
        V8 kn[4], qn[4];
        if (nh < p.heads) load_qk(nh, kn, qn);                      // the next head's Q / K while this one computes
        const T* vrow = vs + (size_t)(head * HD + (l31 & 15)) * VLD;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            if (qt >= nqt) break;                                   // wave-uniform
            // ---- S^T tile by tile: register 4 b + e of lane (query l31, h) is key 32 kt + 8 b + 4 h + e
            f32x16 sc[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt >= nkt) break;
                f32x16 a;
#pragma unroll
                for (int e = 0; e < 16; ++e) a[e] = 0.f;
                a = Mma32<T>::k16(kf[kt], qf[qt], a);
                if (kt * 32 + 32 > nvalid) {                        // the ragged last tile: keys past the real tokens
                    asm volatile("; ragged key tile");
#pragma unroll
                    for (int e = 0; e < 16; ++e) a[e] = kt * 32 + 8 * (e >> 2) + 4 * h + (e & 3) < nvalid ? a[e] : -INFINITY;
                }
                sc[kt] = a;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt >= nkt) break;
                float lm = max3(sc[kt][0], sc[kt][1], sc[kt][2]);
#pragma unroll
                for (int e = 3; e < 15; e += 2) lm = max3(lm, sc[kt][e], sc[kt][e + 1]);
                mx = max3(mx, lm, sc[kt][15]);
            }
            mx = xor32_max(mx);
            // ---- P = exp2(S - max) packed per 16-key chunk as the B operand; O^T = V^T P^T with V^T fragments out of LDS. Rows 16-31
            // of the operand are not channels: row 16 is all ones (the softmax denominator comes out of the same MFMAs), the rest zero
            f32x16 o;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (c >= nch) break;
                V8 pf, vf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = from_f32<T>(__builtin_amdgcn_exp2f(sc[c >> 1][8 * (c & 1) + e] - mx));
                if (l31 >= 16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vf[e] = from_f32<T>(l31 == 16 ? 1.f : 0.f);
                } else {
                    // runs of 4 keys; a run past the end is clamped to the last run (finite values; its P is exactly 0)
                    int k0 = 16 * c + 4 * h, k1 = k0 + 8;
                    k0 = k0 + 4 <= nkeys ? k0 : nkeys - 4; k1 = k1 + 4 <= nkeys ? k1 : nkeys - 4;
                    const V4 lo = *(const V4*)(vrow + k0), hi = *(const V4*)(vrow + k1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                }
                o = Mma32<T>::k16(vf, pf, o);
            }
            // ---- normalise; register 4 b + e of lane (query, h) is channel 8 b + 4 h + e, the ones row is channel 16 = (b 2, h 0, e 0)
            const auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[8]), __float_as_uint(o[8]), false, false);
            const float inv = 1.f / (h == 0 ? o[8] : __uint_as_float(rr[0]));
            V4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = from_f32<T>(o[e] * inv); hi[e] = from_f32<T>(o[4 + e] * inv); }
            const u32x2 lu = __builtin_bit_cast(u32x2, lo), hu = __builtin_bit_cast(u32x2, hi);
            // lower lanes keep lo (channels 0-3) and take the upper lanes' lo (4-7); upper lanes take the lower lanes' hi (8-11) and keep hi
            const auto s0 = __builtin_amdgcn_permlane32_swap(lu[0], hu[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(lu[1], hu[1], false, false);
            const u32x4v st = {s0[0], s1[0], s0[1], s1[1]};
            const int q = qt * 32 + l31;
            if (q < nkeys) *(u32x4v*)(os + (size_t)q * OLD + head * HD + 8 * h) = st;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { kf[t] = kn[t]; qf[t] = qn[t]; }
        head = nh;
    }
    __syncthreads();
    // ---- the window's rows, 16 bytes per lane, consecutive lanes = consecutive pieces (one contiguous run when ldo = C)
    {
        const int cpr = C >> 3, total = nkeys * cpr;
        T* __restrict__ out = (T*)p.out + ((long)b * p.Tp + tok0) * p.ldo;
        for (int i = tid; i < total; i += 256) {
            const int row = i / cpr, c8 = i - row * cpr;
            *(u32x4v*)(out + (long)row * p.ldo + 8 * c8) = *(const u32x4v*)(os + (size_t)row * OLD + 8 * c8);
        }
    }
}

template <typename T>
int launch_wtile(const lwdetr_attn_desc& p, hipStream_t st) {
    const int C = p.heads * 16, VLD = ((p.keys_per_seq + 7) & ~7) + 8;
    const size_t lds = ((size_t)C * VLD + (size_t)p.keys_per_seq * (C + 8)) * sizeof(T);
    static signed char state[16] = {};          // per device: 0 = not asked yet, 1 = granted, -1 = refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return LWDETR_ERR_UNSUPPORTED;
    if (state[dev] == 0)
        state[dev] = hipFuncSetAttribute((const void*)attn_wtile_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess ? 1 : -1;
    if (state[dev] < 0) { (void)hipGetLastError(); return LWDETR_ERR_UNSUPPORTED; }
    const double nseq = (double)p.B * p.seqs_per_img;
    const double flops = 4.0 * nseq * p.heads * (double)p.keys_per_seq * p.keys_per_seq * 16;
    const double bytes = 4.0 * nseq * p.heads * p.keys_per_seq * 16 * sizeof(T);
    const int kid = p.kind == 0 ? KID_ATTN_WINDOW : (p.kind == 1 ? KID_ATTN_GLOBAL : KID_ATTN_DECODER);
    ProfScope ps(kid, flops, bytes, st);
    hipLaunchKernelGGL((attn_wtile_kernel<T>), dim3((unsigned)(p.B * p.seqs_per_img)), dim3(256), lds, st, p);
    return lwdetr_check_launch();
}
#endif

template <typename T, int HD, int QT>
int launch_qt(const lwdetr_attn_desc& p, hipStream_t st) {
    const int units = (p.keys_per_seq + 16 * QT - 1) / (16 * QT);
    dim3 grid((units + 3) / 4, p.heads, p.B * p.seqs_per_img);
    const double nseq = (double)p.B * p.seqs_per_img;
    const double flops = 4.0 * nseq * p.heads * (double)p.keys_per_seq * p.keys_per_seq * HD;
    const double bytes = 4.0 * nseq * p.heads * p.keys_per_seq * HD * sizeof(T);
    const int kid = p.kind == 0 ? KID_ATTN_WINDOW : (p.kind == 1 ? KID_ATTN_GLOBAL : KID_ATTN_DECODER);
    ProfScope ps(kid, flops, bytes, st);
    const bool short_off = lwdetr_knob(KNOB_ATTN_SHORT, 1) == 0;         // LWDETR_ATTN_SHORT=0: the double-buffered form for every length (A-B runs)
    if constexpr (QT == 2 && sizeof(T) == 2) {
        if (p.keys_per_seq <= 128 && !short_off) {
            hipLaunchKernelGGL((attn_kernel<T, HD, QT, true>), grid, dim3(256), 0, st, p);
            return lwdetr_check_launch();
        }
    }
    hipLaunchKernelGGL((attn_kernel<T, HD, QT>), grid, dim3(256), 0, st, p);
    return lwdetr_check_launch();
}

template <typename T, int HD>
int launch(const lwdetr_attn_desc& p, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && HD == 16) {
        // window tiles through LDS (round 5): one workgroup per (image, window), all heads. Built for the north star's "LDS-staged window
        // tiles" and measured SLOWER than the per-head waves below (tools/attn_bench.py, hd 16, 100-key windows, us per launch incl. ~5 us
        // of launch overhead: B = 32 attn_kernel 33.7 | one wave per (window, head) 38.6 | this kernel 61.1; B = 16 19.3 | 25.1 | 34.9;
        // config 2 13.4-13.5 k -> 13.0 k img/s), and the ablations of the one-wave kernel say why staging cannot pay here: with EVERY
        // load and store removed it still runs 29.6 of its 38.6 us, without the exponentials 37.5 - window attention at hd 16 is bound by
        // the dependent MFMA -> max -> exp -> MFMA chain of a (window, head) at 2-3 waves per SIMD, not by its 79 MB of traffic
        // (profiles/r5c_window_attention_hd16.txt). Off unless asked for: LWDETR_ATTN_WTILE=1 (tests keep it bit-identical to the
        // one-wave kernel).
#ifdef LWDETR_EXPERIMENTS
        const int wmode = (int)lwdetr_knob(KNOB_ATTN_WTILE, 0);
        const int C = p.heads * 16, VLD = ((p.keys_per_seq + 7) & ~7) + 8;
        const size_t lds = ((size_t)C * VLD + (size_t)p.keys_per_seq * (C + 8)) * sizeof(T);
        // (keys % 4 / Tp % 4: the V^T staging moves 4-key pieces and clamps at keys - 4 - advisor r5)
        if (wmode != 0 && p.keys_per_seq <= 128 && p.keys_per_seq >= 4 && p.keys_per_seq % 4 == 0 && p.Tp % 4 == 0 &&
            p.sub_stride >= p.keys_per_seq && p.ldo % 8 == 0 && ((size_t)p.out & 15) == 0 &&
            ((size_t)p.VT & 7) == 0 && p.seq_tok_stride % 4 == 0 && lds <= 128 * 1024) {
            const int rc = launch_wtile<T>(p, st);
            if (rc != LWDETR_ERR_UNSUPPORTED) return rc;
        }
#endif
    }
    if constexpr (sizeof(T) == 2 && (HD == 16 || HD == 32)) {
        // one wave per (sequence, head): sequences of at most 128 keys whose pad rows sit behind the real tokens.
        // Measured (tools/attn_bench.py, us per launch, attn_kernel | this kernel): hd 32, 100-key windows, medium B = 64 bf16
        // 108 | 82, large B = 32 fp16 51 | 41; hd 16 (small, B = 16 / 32): 20.4 | 22.0, 32.8 | 36.1 - both forms run their loads
        // and their exp-bound arithmetic in lockstep there and the four-wave form spreads the arithmetic wider.
        // LWDETR_ATTN_WIN: 0 = never, 1 = whenever legal (tests), default = hd 32 only.
        const int mode = (int)lwdetr_knob(KNOB_ATTN_WIN, 2);
        if (mode != 0 && (mode == 1 || HD == 32) && p.keys_per_seq <= 128 && p.sub_stride >= p.keys_per_seq && p.ldo % 8 == 0 &&
            ((size_t)p.out & 15) == 0 && (long)p.B * p.heads * p.seqs_per_img / 4 < 0x7fffffffL)
            return launch_win<T, HD>(p, st);
    }
    if (LdsPath<T, HD>::ok(p) && lds_path_applies(p)) return LdsPath<T, HD>::go(p, st);
    // long sequences: 64 queries per wave (K / V^T fragments amortised over 4 query tiles); short ones keep 32 so a
    // 100-token window still spreads over 4 waves
    // ... and only when that still leaves at least one workgroup per CU (a single image has 84 of them at 64 queries per
    // wave: 25 us per launch against 18 us with 32)
    const long wgs4 = (long)((p.keys_per_seq + 255) / 256) * p.heads * p.B * p.seqs_per_img;
    const bool qt4 = lwdetr_knob_is_set(KNOB_ATTN_QT) ? lwdetr_knob(KNOB_ATTN_QT, 0) == 4 : (p.keys_per_seq >= 1024 && HD <= 32 && sizeof(T) == 2 && wgs4 >= 256);
    if (qt4) return launch_qt<T, HD, (HD <= 32 ? 4 : 2)>(p, st);
    return launch_qt<T, HD, 2>(p, st);
}

template <typename T>
int dispatch_hd(const lwdetr_attn_desc& p, hipStream_t st) {
    switch (p.hd) {
        case 16: return launch<T, 16>(p, st);
        case 32: return launch<T, 32>(p, st);
        case 64: return launch<T, 64>(p, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

}  // namespace

#ifdef LWDETR_ATTN_HD16_TU
// (this object: the hd 16 ring kernels only; `cfg` = the tuning override of the main object, 0 = none)
extern "C" __attribute__((visibility("hidden"))) int lwdetr_attn_lds_hd16_impl(const lwdetr_attn_desc* p, int dtype, void* hip_stream, int cfg) {
    g_attn_lds_cfg = cfg;
    return dtype == DT_F16 ? launch_lds_cfg<f16, 16>(*p, (hipStream_t)hip_stream) : launch_lds_cfg<bf16, 16>(*p, (hipStream_t)hip_stream);
}
#else
extern "C" void lwdetr_attention_tuning(int lds_mode) { g_attn_lds_mode = lds_mode; }
extern "C" void lwdetr_attention_tuning_cfg(int cfg) { g_attn_lds_cfg = cfg; }

extern "C" int lwdetr_attention(const lwdetr_attn_desc* desc, int dtype, void* hip_stream) {
    if (!desc) return LWDETR_ERR_BAD_ARG;
    const lwdetr_attn_desc& p = *desc;
    if (!p.Q || !p.K || !p.VT || !p.out || p.B <= 0 || p.heads <= 0 || p.Tp <= 0 || p.Tp % 4 != 0) return LWDETR_ERR_BAD_ARG;
    if (p.seqs_per_img <= 0 || p.keys_per_seq < 8 || p.keys_per_seq % 4 != 0 || p.sub_stride <= 0 || p.sub_len <= 0 || p.sub_len > p.sub_stride)
        return LWDETR_ERR_BAD_ARG;
    if ((long)(p.seqs_per_img - 1) * p.seq_tok_stride + p.keys_per_seq > p.Tp) return LWDETR_ERR_BAD_ARG;
    if (p.seq_tok_stride % 4 != 0 || p.ldo % 4 != 0) return LWDETR_ERR_BAD_ARG;
    if ((long)p.B * p.seqs_per_img > 65535 || p.heads > 65535) return LWDETR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F32: return dispatch_hd<float>(p, st);
        case DT_F16: return dispatch_hd<f16>(p, st);
        case DT_BF16: return dispatch_hd<bf16>(p, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}
#endif  // LWDETR_ATTN_HD16_TU
