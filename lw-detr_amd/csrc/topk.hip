// Sorted top-k selection for gfx950, one 1024-thread workgroup per image, and the two places LW-DETR uses it:
//   * two-stage query selection: class-max of the encoder logits (rowmax kernel) -> top-nq rows in descending order
//     (reference models/transformer.py:246-248, torch.topk(enc_outputs_class.max(-1)[0], num_queries, dim=1))
//   * PostProcess: top-num_select of sigmoid(logits) over nq x classes, labels = idx % C, boxes = idx // C gathered,
//     cxcywh -> xyxy, scaled by the target size (reference models/lwdetr.py:509-540, util/box_ops.py:21-25)
// Selection = MSB-first radix select (11-bit digits, LDS histograms) on a 52-bit composite
//     comp = order_preserving_u32(value) << 20 | (0xFFFFF - index)
// which is unique per element, so the selected SET and its ORDER are fully deterministic: descending value, equal
// values by ascending index (torch.topk leaves the order of ties unspecified). The select stops as soon as the bin
// holding the k-th element is taken whole (3 passes when the k-th value is not tied). The k winners are then ranked
// against each other in LDS (k*k/1024 compares per thread) instead of sorted.
#include "common.h"

namespace {

constexpr int TK_THREADS = 1024;
constexpr int TK_MAXK = 1024;
constexpr int TK_IDX_BITS = 20;
constexpr unsigned TK_IDX_MASK = (1u << TK_IDX_BITS) - 1;

__device__ __forceinline__ unsigned fkey(float f) {          // monotone float -> u32 (ascending)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

__device__ __forceinline__ unsigned long long tk_comp(unsigned key, int i) {
    return ((unsigned long long)key << TK_IDX_BITS) | (unsigned long long)(TK_IDX_MASK - (unsigned)i);
}

struct TopkShared {
    unsigned hist[2048];
    unsigned long long sel[TK_MAXK];
    unsigned rank[TK_MAXK];
    unsigned bstar, above, cnt, nsel;
};

// Leaves the k winners' composites in sh.sel[0..k) and their descending ranks in sh.rank[0..k). All threads call it.
// CACHE: the 32-bit keys of the image are written to LDS (`keys`, N words of dynamic shared memory) by the first pass and
// read from there by the later ones - one dependent L2 round trip per element instead of four or five.
template <typename T, bool CACHE>
__device__ void topk_block(const T* __restrict__ x, int N, int K, TopkShared& sh, unsigned* __restrict__ keys) {
    const int tid = threadIdx.x;
    unsigned long long prefix = 0;          // digits decided so far (the high bits of the threshold composite)
    unsigned rem = (unsigned)K;             // how many elements are still to be taken among those matching the prefix
    unsigned long long thr = 0;
    int top = 32 + TK_IDX_BITS;             // bits not yet decided
    bool first = true;
    while (true) {
        const int bits = top > 30 ? 11 : 10;            // 52 = 11 + 11 + 10 + 10 + 10
        const int shift = top - bits;
        const unsigned nb = 1u << bits;
        for (int i = tid; i < 2048; i += TK_THREADS) sh.hist[i] = 0;
        __syncthreads();
        if (first) {
            // the only pass that touches global memory (CACHE): 4 independent loads in flight per thread
            int i = tid;
            for (; i + 3 * TK_THREADS < N; i += 4 * TK_THREADS) {
                unsigned k4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) k4[u] = fkey(to_f32<T>(x[i + u * TK_THREADS]));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (CACHE) keys[i + u * TK_THREADS] = k4[u];
                    atomicAdd(&sh.hist[k4[u] >> 21], 1u);           // first digit = the top 11 of the 32 key bits
                }
            }
            for (; i < N; i += TK_THREADS) {
                const unsigned k = fkey(to_f32<T>(x[i]));
                if (CACHE) keys[i] = k;
                atomicAdd(&sh.hist[k >> 21], 1u);
            }
        } else {
            for (int i = tid; i < N; i += TK_THREADS) {
                const unsigned long long c = tk_comp(CACHE ? keys[i] : fkey(to_f32<T>(x[i])), i);
                if ((c >> top) == prefix) atomicAdd(&sh.hist[(unsigned)(c >> shift) & (nb - 1)], 1u);
            }
        }
        first = false;
        __syncthreads();
        if (tid < 64) {                                  // wave 0: lane l owns bins [l * per, (l + 1) * per)
            const int per = (int)nb >> 6;
            unsigned own = 0;
            for (int j = 0; j < per; ++j) own += sh.hist[tid * per + ((j + tid) & (per - 1))];
            unsigned s = own;                            // inclusive suffix sum over lanes (lane 63 = highest bins)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_down(s, o);
                if (tid + o < 64) s += t;
            }
            const unsigned excl = s - own;
            if (excl < rem && rem <= s) {
                unsigned acc = excl;
                for (int j = per - 1; j >= 0; --j) {
                    const unsigned c = sh.hist[tid * per + j];
                    if (acc + c >= rem) { sh.bstar = (unsigned)(tid * per + j); sh.above = acc; sh.cnt = c; break; }
                    acc += c;
                }
            }
        }
        __syncthreads();
        const unsigned bstar = sh.bstar, above = sh.above, cnt = sh.cnt;
        rem -= above;
        prefix = (prefix << bits) | bstar;
        top = shift;
        if (cnt == rem || top == 0) { thr = prefix << top; break; }     // the whole bin is taken: threshold found
    }
    if (tid == 0) sh.nsel = 0;
    for (int i = tid; i < TK_MAXK; i += TK_THREADS) sh.rank[i] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += TK_THREADS) {
        const unsigned long long c = tk_comp(CACHE ? keys[i] : fkey(to_f32<T>(x[i])), i);
        if (c >= thr) {
            const unsigned p = atomicAdd(&sh.nsel, 1u);
            if (p < (unsigned)K) sh.sel[p] = c;
        }
    }
    __syncthreads();
    // rank = number of winners with a larger composite; `parts` threads share one winner's k compares
    const int parts = TK_THREADS / K > 0 ? TK_THREADS / K : 1;
    const int e = tid % K, p = tid / K;
    if (p < parts) {
        const unsigned long long mine = sh.sel[e];
        const int j0 = (int)((long)K * p / parts), j1 = (int)((long)K * (p + 1) / parts);
        unsigned r = 0;
        for (int j = j0; j < j1; ++j) r += sh.sel[j] > mine ? 1u : 0u;
        atomicAdd(&sh.rank[e], r);
    }
    __syncthreads();
}

template <typename T, bool CACHE>
__global__ __launch_bounds__(TK_THREADS) void topk_kernel(const T* __restrict__ x, int N, int K, int64_t* __restrict__ idx_out,
                                                          float* __restrict__ val_out) {
    __shared__ TopkShared sh;
    extern __shared__ unsigned tk_keys[];
    const int b = blockIdx.x;
    topk_block<T, CACHE>(x + (long)b * N, N, K, sh, tk_keys);
    const int tid = threadIdx.x;
    if (tid < K) {
        const unsigned long long c = sh.sel[tid];
        const unsigned r = sh.rank[tid];
        idx_out[(long)b * K + r] = (int64_t)(TK_IDX_MASK - (unsigned)(c & TK_IDX_MASK));
        if (val_out) val_out[(long)b * K + r] = fkey_inv((unsigned)(c >> TK_IDX_BITS));
    }
}

// PostProcess: sigmoid is monotone, so the top-k runs on the logits and only the k winners go through sigmoid.
template <typename T, bool CACHE>
__global__ __launch_bounds__(TK_THREADS) void postprocess_kernel(const T* __restrict__ logits, const T* __restrict__ boxes,
                                                                 const float* __restrict__ sizes, int nq, int ncls, int K,
                                                                 float* __restrict__ scores, int64_t* __restrict__ labels,
                                                                 float* __restrict__ out_boxes, float* __restrict__ packed) {
    __shared__ TopkShared sh;
    extern __shared__ unsigned tk_keys[];
    const int b = blockIdx.x;
    const int N = nq * ncls;
    topk_block<T, CACHE>(logits + (long)b * N, N, K, sh, tk_keys);
    const int tid = threadIdx.x;
    if (tid < K) {
        const unsigned long long c = sh.sel[tid];
        const long o = (long)b * K + sh.rank[tid];
        const int i = (int)(TK_IDX_MASK - (unsigned)(c & TK_IDX_MASK));
        const int q = i / ncls;
        const float v = fkey_inv((unsigned)(c >> TK_IDX_BITS));
        const float score = to_f32<T>(from_f32<T>(1.f / (1.f + expf(-v))));   // sigmoid evaluated in f32, stored at T's precision
        const T* bx = boxes + ((long)b * nq + q) * 4;
        const float cx = to_f32<T>(bx[0]), cy = to_f32<T>(bx[1]);
        const float w = fmaxf(to_f32<T>(bx[2]), 0.f), h = fmaxf(to_f32<T>(bx[3]), 0.f);
        const float ih = sizes[b * 2], iw = sizes[b * 2 + 1];               // target_sizes rows are (h, w)
        // corners are formed at T's precision (the reference computes them on the model-dtype tensor) and scaled in f32
        const float x0 = to_f32<T>(from_f32<T>(cx - 0.5f * w)) * iw, y0 = to_f32<T>(from_f32<T>(cy - 0.5f * h)) * ih;
        const float x1 = to_f32<T>(from_f32<T>(cx + 0.5f * w)) * iw, y1 = to_f32<T>(from_f32<T>(cy + 0.5f * h)) * ih;
        if (packed) {       // (B, K, 6) f32 rows: score, label, x0, y0, x1, y1 - the record the detection all-gather ships
            float* pr = packed + o * 6;
            pr[0] = score; pr[1] = (float)(i - q * ncls); pr[2] = x0; pr[3] = y0; pr[4] = x1; pr[5] = y1;
        } else {
            scores[o] = score;
            labels[o] = (int64_t)(i - q * ncls);
            out_boxes[o * 4 + 0] = x0; out_boxes[o * 4 + 1] = y0; out_boxes[o * 4 + 2] = x1; out_boxes[o * 4 + 3] = y1;
        }
    }
}

// Row maximum over the first `ncols` entries of each row: 16 lanes per row, 4-element vector loads (ld % 4 == 0).
template <typename T>
__global__ __launch_bounds__(256) void rowmax_kernel(const T* __restrict__ x, long ld, long rows, int ncols, float* __restrict__ out) {
    constexpr int EPC = 4;
    typedef T VC __attribute__((ext_vector_type(EPC)));
    const int lane16 = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    float m = -INFINITY;
    if (row < rows) {
        const T* xr = x + row * ld;
        const int full = ncols / EPC;
        for (int c = lane16; c < full; c += 16) {
            const VC t = *(const VC*)(xr + c * EPC);
#pragma unroll
            for (int e = 0; e < EPC; ++e) m = fmaxf(m, to_f32<T>(t[e]));
        }
        for (int c = full * EPC + lane16; c < ncols; c += 16) m = fmaxf(m, to_f32<T>(xr[c]));
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (row < rows && lane16 == 0) out[row] = m;
}

}  // namespace

// key cache: whatever is left of the 160 KB beside TopkShared; larger inputs (e.g. 300 x 366 Objects365 logits) re-read global
constexpr size_t TK_CACHE_BYTES = 160 * 1024 - sizeof(TopkShared) - 256;
static bool tk_allow_lds(const void* fn) {
    // once per kernel instantiation (keyed by its address); hipFuncSetAttribute is not a stream operation
    static const void* done[16]; static int ndone = 0;
    for (int i = 0; i < ndone; ++i) if (done[i] == fn) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TK_CACHE_BYTES) != hipSuccess) return false;
    if (ndone < 16) done[ndone++] = fn;
    return true;
}

#define LWDETR_DISPATCH_T(dtype, CALL)                 \
    switch (dtype) {                                   \
        case DT_F32: { typedef float TT; CALL; break; } \
        case DT_F16: { typedef f16 TT; CALL; break; }   \
        case DT_BF16: { typedef bf16 TT; CALL; break; } \
        default: return LWDETR_ERR_UNSUPPORTED;        \
    }

extern "C" int lwdetr_rowmax(const void* x, long ld, long rows, int ncols, float* out, int dtype, void* hip_stream) {
    if (!x || !out || rows < 0 || ncols <= 0 || ld < ncols) return LWDETR_ERR_BAD_ARG;
    if (rows == 0) return LWDETR_OK;
    if (ld % 4 || ((uintptr_t)x & 15)) return LWDETR_ERR_BAD_ARG;                // vector-aligned row starts
    hipStream_t st = (hipStream_t)hip_stream;
    ProfScope ps(KID_ELTWISE, 0.0, (double)rows * ncols * (dtype == DT_F32 ? 4 : 2), st);
    LWDETR_DISPATCH_T(dtype, hipLaunchKernelGGL((rowmax_kernel<TT>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st,
                                                (const TT*)x, ld, rows, ncols, out));
    return lwdetr_check_launch();
}

extern "C" int lwdetr_topk(const void* x, int B, int N, int K, int64_t* idx_out, float* val_out, int dtype, void* hip_stream) {
    if (!x || !idx_out || B < 0 || K <= 0 || K > N || K > TK_MAXK || N > (int)TK_IDX_MASK) return LWDETR_ERR_BAD_ARG;
    if (B == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    ProfScope ps(KID_ELTWISE, 0.0, (double)B * N * (dtype == DT_F32 ? 4 : 2), st);
    const bool cache = (size_t)N * 4 <= TK_CACHE_BYTES;
    LWDETR_DISPATCH_T(dtype, {
        if (cache) {
            if (!tk_allow_lds((const void*)topk_kernel<TT, true>)) return LWDETR_ERR_LAUNCH;
            hipLaunchKernelGGL((topk_kernel<TT, true>), dim3(B), dim3(TK_THREADS), (size_t)N * 4, st, (const TT*)x, N, K, idx_out, val_out);
        } else {
            hipLaunchKernelGGL((topk_kernel<TT, false>), dim3(B), dim3(TK_THREADS), 0, st, (const TT*)x, N, K, idx_out, val_out);
        }
    });
    return lwdetr_check_launch();
}

static int postprocess_impl(const void* logits, const void* boxes, const float* target_sizes, int B, int nq, int ncls, int K,
                                  float* scores, int64_t* labels, float* out_boxes, float* packed, int dtype, void* hip_stream) {
    if (!logits || !boxes || !target_sizes || B < 0 || nq <= 0 || ncls <= 0 || K <= 0 ||
        K > TK_MAXK || (long)nq * ncls > (long)TK_IDX_MASK || K > nq * ncls)
        return LWDETR_ERR_BAD_ARG;
    if (B == 0) return LWDETR_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    ProfScope ps(KID_ELTWISE, 0.0, (double)B * nq * ncls * (dtype == DT_F32 ? 4 : 2), st);
    const size_t kb = (size_t)nq * ncls * 4;
    const bool cache = kb <= TK_CACHE_BYTES;
    LWDETR_DISPATCH_T(dtype, {
        if (cache) {
            if (!tk_allow_lds((const void*)postprocess_kernel<TT, true>)) return LWDETR_ERR_LAUNCH;
            hipLaunchKernelGGL((postprocess_kernel<TT, true>), dim3(B), dim3(TK_THREADS), kb, st, (const TT*)logits, (const TT*)boxes,
                               target_sizes, nq, ncls, K, scores, labels, out_boxes, packed);
        } else {
            hipLaunchKernelGGL((postprocess_kernel<TT, false>), dim3(B), dim3(TK_THREADS), 0, st, (const TT*)logits, (const TT*)boxes,
                               target_sizes, nq, ncls, K, scores, labels, out_boxes, packed);
        }
    });
    return lwdetr_check_launch();
}

extern "C" int lwdetr_postprocess(const void* logits, const void* boxes, const float* target_sizes, int B, int nq, int ncls, int K,
                                  float* scores, int64_t* labels, float* out_boxes, int dtype, void* hip_stream) {
    if (!scores || !labels || !out_boxes) return LWDETR_ERR_BAD_ARG;
    return postprocess_impl(logits, boxes, target_sizes, B, nq, ncls, K, scores, labels, out_boxes, nullptr, dtype, hip_stream);
}

extern "C" int lwdetr_postprocess_packed(const void* logits, const void* boxes, const float* target_sizes, int B, int nq, int ncls,
                                         int K, float* packed, int dtype, void* hip_stream) {
    if (!packed) return LWDETR_ERR_BAD_ARG;
    return postprocess_impl(logits, boxes, target_sizes, B, nq, ncls, K, nullptr, nullptr, nullptr, packed, dtype, hip_stream);
}
