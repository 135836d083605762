// Shared device/host helpers for the gfx950 (CDNA4) kernels of lwdetr_amd.
// Wave = 64 lanes; MFMA fragments follow the 16x16 family layouts:
//   A operand: lane l holds A[i = l & 15][k-run selected by g = l >> 4]
//   B operand: lane l holds B[k-run selected by g][j = l & 15]
//   C/D      : lane l holds D[i = 4 * g + r][j = l & 15], r = 0..3
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lwdetr_hip.h"   // public C ABI (include/), shared POD descriptors

enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Vec;
template <> struct Vec<float> { typedef f32x4 v4; typedef f32x8 v8; };
template <> struct Vec<f16> { typedef f16x4 v4; typedef f16x8 v8; };
template <> struct Vec<bf16> { typedef bf16x4 v4; typedef bf16x8 v8; };

template <typename T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }

// ---- MFMA wrappers: one call contracts 16 (k16) or 32 (k32) k-values held as 4 / 8 consecutive elements per lane.
// For f32 the contraction is issued as 4 / 8 exact-f32 16x16x4 MFMAs (element s of every lane forms k-slice s;
// the k order inside the chunk is a permutation shared by both operands, which leaves the product unchanged).
template <typename T> struct Mma;
template <> struct Mma<f16> {
    static __device__ __forceinline__ f32x4 k32(f16x8 a, f16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 k16(f16x4 a, f16x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<bf16> {
    static __device__ __forceinline__ f32x4 k32(bf16x8 a, bf16x8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 k16(bf16x4 a, bf16x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b),
                                                         c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ f32x4 k32(f32x8 a, f32x8 b, f32x4 c) {
#pragma unroll
        for (int s = 0; s < 8; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
        return c;
    }
    static __device__ __forceinline__ f32x4 k16(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], c, 0, 0, 0);
        return c;
    }
};

template <typename T> __device__ __forceinline__ typename Vec<T>::v4 cvt4(f32x4 v) {
    typename Vec<T>::v4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = from_f32<T>(v[i]);
    return o;
}
template <typename T> __device__ __forceinline__ f32x4 up4(typename Vec<T>::v4 v) {
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = to_f32<T>(v[i]);
    return o;
}

// ---- activations (match torch: exact-erf GELU, SiLU = x * sigmoid(x))
enum { ACT_NONE = LWDETR_ACT_NONE, ACT_RELU = LWDETR_ACT_RELU, ACT_GELU = LWDETR_ACT_GELU, ACT_SILU = LWDETR_ACT_SILU };
__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.f ? x : 0.f;
        case ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
        case ACT_SILU: return x / (1.f + __expf(-x));
        default: return x;
    }
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): GELU stays exact-erf to f32 round-off at a fraction of erff's cost
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.f - poly * __expf(-ax * ax);
    return x < 0.f ? -e : e;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + fast_erf(x * 0.70710678118654752440f)); }

// GELU for 16-bit storage types: x * sigmoid(x * (c0 + c1 x^2 + c2 x^4)), coefficients fitted (minimax on [-9, 9]) to the
// exact erf form: max |error| 2.5e-5, i.e. well below one f16 / bf16 ulp of the result; 9 VALU ops instead of ~20.
__device__ __forceinline__ float gelu_fast16(float x) {
    const float x2 = fminf(x * x, 36.f);                       // beyond |x| = 6 the result is x or 0 to < 1e-8
    const float p = fmaf(fmaf(x2, 0.0010142630555f, -0.1067757240036f), x2, -2.3011213394584f);   // -(c0 + c1 x2 + c2 x2^2) * log2(e)
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * p));
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) { return gelu_fast16(x); }
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }

// ---- token layouts. A "row" m of an activation matrix addresses one token of one image.
//  RASTER   : m = (b * Hp + y) * Wp + x
//  WINMAJOR : m = b * Tp + win * Twp + i,  win = (y / h) * 4 + (x / w), i = (y % h) * w + (x % w), h = Hp/4, w = Wp/4,
//             Twp = h*w rounded up to a multiple of 4 (pad rows i >= h*w carry no token), Tp = 16 * Twp.
// The ViT keeps tokens window-major for its whole depth (reference: models/backbone/vit.py:353-358).
typedef lwdetr_tok_layout TokLayout;   // {winmajor, Hp, Wp, Twp}
struct TokPos { int b, y, x, valid; };

__device__ __forceinline__ TokPos tok_decode(long m, const TokLayout& L) {
    TokPos p;
    if (!L.winmajor) {
        const int hw = L.Hp * L.Wp;
        p.b = (int)(m / hw);
        const int r = (int)(m - (long)p.b * hw);
        p.y = r / L.Wp; p.x = r - p.y * L.Wp; p.valid = 1;
    } else {
        const int Tp = 16 * L.Twp, h = L.Hp >> 2, w = L.Wp >> 2;
        p.b = (int)(m / Tp);
        const int r = (int)(m - (long)p.b * Tp);
        const int win = r / L.Twp, i = r - win * L.Twp;
        const int iy = i / w, ix = i - iy * w;
        p.valid = i < h * w;
        p.y = (win >> 2) * h + iy; p.x = (win & 3) * w + ix;
    }
    return p;
}
__device__ __forceinline__ long tok_encode(int b, int y, int x, const TokLayout& L) {
    if (!L.winmajor) return ((long)b * L.Hp + y) * L.Wp + x;
    const int h = L.Hp >> 2, w = L.Wp >> 2;
    const int wy = y / h, wx = x / w;
    return (long)b * 16 * L.Twp + (wy * 4 + wx) * L.Twp + (y - wy * h) * w + (x - wx * w);
}

// ---- profiling hooks (prof.hip): per-kernel HIP-event timing on the launch stream, off by default.
enum {
    KID_MSDA = 0, KID_MSDA_GENERIC, KID_MSDA_FUSED, KID_GEMM, KID_GEMM_CONV, KID_GEMM_PATCH, KID_ATTN_WINDOW,
    KID_ATTN_GLOBAL, KID_ATTN_DECODER, KID_LAYERNORM, KID_ELTWISE, KID_MLP, KID_VITBLOCK, KID_CHAIN, KID_COUNT
};
void lwdetr_prof_begin(int kid, double flops, double bytes, hipStream_t s);
void lwdetr_prof_end(hipStream_t s);
struct ProfScope {
    hipStream_t s;
    ProfScope(int kid, double flops, double bytes, hipStream_t st) : s(st) { lwdetr_prof_begin(kid, flops, bytes, st); }
    ~ProfScope() { lwdetr_prof_end(s); }
};

// ---- run-time switches of the launch paths (tuning, A/B runs, tests). Every LWDETR_* variable a launch path looks at is read from the
// environment ONCE, when the library first needs any of them (prof.hip), into this table - no getenv on a launch path; tests and tools that
// switch inside one process go through lwdetr_tuning_set (C ABI). A knob that is not set returns the caller's default.
enum {
    KNOB_ATTN_LDS_CFG = 0, KNOB_ATTN_LDS, KNOB_ATTN_SHORT, KNOB_ATTN_WTILE, KNOB_ATTN_WIN, KNOB_ATTN_QT, KNOB_CHAIN_SPLIT_ROWS, KNOB_GEMM_BIG,
    KNOB_GEMM_BIG_BN, KNOB_GEMM_BIG_2WG, KNOB_CONV_PATCH, KNOB_GEMM_TILE, KNOB_GEMM_DMA, KNOB_GEMM_KB, KNOB_GEMM_NST, KNOB_GEMM_PT,
    KNOB_GEMM_PT_SKEW, KNOB_MLP_SMALL_TT, KNOB_FFN_SPLITS, KNOB_MLP_SMALL, KNOB_VB_GRID, KNOB_VB_GELU16, KNOB_VB_HALF, KNOB_GEMM_FEW_WAVES, KNOB_COUNT
};
long lwdetr_knob(int id, long dflt);
bool lwdetr_knob_is_set(int id);

static inline int lwdetr_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LWDETR_OK : LWDETR_ERR_LAUNCH;
}
