// Multi-scale deformable attention, forward, for gfx950.
//
// Replaces (reference, /root/reference/models/ops/src): cuda/ms_deform_im2col_cuda.cuh:33-84 (bilinear helper),
// :237-299 (forward kernel), cuda/ms_deform_attn_cuda.cu:20-80 (host wrapper incl. im2col_step chunking, which a
// single launch over the whole batch makes unnecessary).  Three entry points:
//   * vec8 kernel   : D % 8 == 0 (the model has D = 16). One lane owns 8 channels of one (b, q, head): every corner
//                     fetch is a 16-byte (f16/bf16) or 32-byte (f32) load, all 4*P corner loads of a level are issued
//                     before any is consumed (the op is gather-latency bound, not FLOP bound), output stores are
//                     fully coalesced, and workgroups of one image are pinned to one XCD so that image's value map
//                     stays in that XCD's L2.
//   * generic kernel: any D / f64 - one thread per output element, for the op-level API contract.
//   * fused kernel  : model path - takes the raw sampling_offsets/attention_weights Linear output and the reference
//                     boxes, and does softmax + location arithmetic (ms_deform_attn.py:117-131) in the prologue.
#include "common.h"

namespace {

struct MsdaParams {
    const void* value; const int64_t* shapes; const int64_t* lsi; const void* loc; const void* aw; void* out;
    int B, S, M, D, L, Q, P;
    int chunks_per_img, xcd_remap;
    // fused variant
    const void* oa; long ld_oa; const float* ref; const float* vr; int oa_logit_off;
};

template <typename T> struct Acc { typedef float t; };
template <> struct Acc<double> { typedef double t; };

template <typename T, typename A> __device__ __forceinline__ A ld(const T* p) { return (A)(*p); }

// ---------------------------------------------------------------------------------------------- generic
template <typename T>
__global__ __launch_bounds__(256) void msda_generic_kernel(MsdaParams p) {
    typedef typename Acc<T>::t A;
    const long n = (long)p.B * p.Q * p.M * p.D;
    const T* value = (const T*)p.value; const T* loc = (const T*)p.loc; const T* aw = (const T*)p.aw;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % p.D);
        const long s_idx = idx / p.D;                       // (b, q, m)
        const int m = (int)(s_idx % p.M);
        const int b = (int)(s_idx / ((long)p.M * p.Q));
        long wp = s_idx * p.L * p.P, lp = wp * 2;
        const long ws = (long)p.M * p.D;
        A col = 0;
        for (int l = 0; l < p.L; ++l) {
            const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
            const T* v = value + ((long)b * p.S + p.lsi[l]) * ws + (long)m * p.D + c;
            for (int pt = 0; pt < p.P; ++pt, ++wp, lp += 2) {
                const A w_im = (A)loc[lp] * W - (A)0.5, h_im = (A)loc[lp + 1] * H - (A)0.5;
                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                    const int hl = (int)floor(h_im), wl = (int)floor(w_im);
                    const A lh = h_im - hl, lw = w_im - wl, hh = 1 - lh, hw = 1 - lw;
                    A v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                    if (hl >= 0 && wl >= 0) v1 = (A)v[((long)hl * W + wl) * ws];
                    if (hl >= 0 && wl + 1 <= W - 1) v2 = (A)v[((long)hl * W + wl + 1) * ws];
                    if (hl + 1 <= H - 1 && wl >= 0) v3 = (A)v[((long)(hl + 1) * W + wl) * ws];
                    if (hl + 1 <= H - 1 && wl + 1 <= W - 1) v4 = (A)v[((long)(hl + 1) * W + wl + 1) * ws];
                    col += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * (A)aw[wp];
                }
            }
        }
        ((T*)p.out)[idx] = (T)col;
    }
}

// ------------------------------------------------------------------------------------------------- vec8
template <typename T> struct Chan8 {          // 8 consecutive channels of one pixel
    typename Vec<T>::v8 v;
    __device__ __forceinline__ void load(const T* p) { v = *(const typename Vec<T>::v8*)p; }
};

// One sampling point: 4 corner offsets (in elements, already clamped in-bounds) + 4 corner weights
// (zeroed for corners / samples outside the map; weight already multiplied by the attention weight).
struct Corners { long o[4]; float w[4]; };

__device__ __forceinline__ Corners make_corners(float loc_x, float loc_y, float attn, int H, int W, long ws) {
    Corners c;
    const float h_im = loc_y * H - 0.5f, w_im = loc_x * W - 0.5f;
    const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
    int hl = inside ? (int)hf : 0, wl = inside ? (int)wf : 0;
    const int hh_i = hl + 1, wh_i = wl + 1;
    const bool t = hl >= 0, l = wl >= 0, bo = hh_i <= H - 1, r = wh_i <= W - 1;
    const int h0 = t ? hl : 0, w0 = l ? wl : 0, h1 = bo ? hh_i : H - 1, w1 = r ? wh_i : W - 1;
    const float a = inside ? attn : 0.f;
    c.w[0] = (t && l) ? hh * hw * a : 0.f;  c.o[0] = ((long)h0 * W + w0) * ws;
    c.w[1] = (t && r) ? hh * lw * a : 0.f;  c.o[1] = ((long)h0 * W + w1) * ws;
    c.w[2] = (bo && l) ? lh * hw * a : 0.f; c.o[2] = ((long)h1 * W + w0) * ws;
    c.w[3] = (bo && r) ? lh * lw * a : 0.f; c.o[3] = ((long)h1 * W + w1) * ws;
    return c;
}

template <typename T, int TP>
__device__ __forceinline__ void sample_level(const T* vbase, const Corners* cs, int P, float (&acc)[8]) {
    constexpr int PP = TP > 0 ? TP : 1;
    if (TP > 0) {
        Chan8<T> v[PP][4];
#pragma unroll
        for (int pt = 0; pt < PP; ++pt)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[pt][k].load(vbase + cs[pt].o[k]);
#pragma unroll
        for (int pt = 0; pt < PP; ++pt)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += cs[pt].w[k] * to_f32<T>(v[pt][k].v[e]);
    } else {
        for (int pt = 0; pt < P; ++pt) {
            Chan8<T> v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k].load(vbase + cs[pt].o[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += cs[pt].w[k] * to_f32<T>(v[k].v[e]);
        }
    }
}

__device__ __forceinline__ bool msda_work_item(const MsdaParams& p, int& b, int& q, int& m, int& dc) {
    const int DC = p.D >> 3;
    const long per_img = (long)p.Q * p.M * DC;
    int img, chunk;
    const int bid = blockIdx.x;
    if (p.xcd_remap) {                       // block b runs on XCD b % 8: keep one image's blocks on one XCD
        const int xcd = bid & 7, slot = bid >> 3;
        img = (slot / p.chunks_per_img) * 8 + xcd; chunk = slot % p.chunks_per_img;
    } else {
        img = bid / p.chunks_per_img; chunk = bid % p.chunks_per_img;
    }
    const long it = (long)chunk * 256 + threadIdx.x;
    if (it >= per_img || img >= p.B) return false;
    dc = (int)(it % DC);
    const long qm = it / DC;
    m = (int)(qm % p.M); q = (int)(qm / p.M); b = img;
    return true;
}

constexpr int MAX_P = 8;

template <typename T, int TP>
__global__ __launch_bounds__(256) void msda_vec8_kernel(MsdaParams p) {
    int b, q, m, dc;
    if (!msda_work_item(p, b, q, m, dc)) return;
    const T* value = (const T*)p.value; const T* loc = (const T*)p.loc; const T* aw = (const T*)p.aw;
    const long ws = (long)p.M * p.D;
    const long sidx = ((long)b * p.Q + q) * p.M + m;
    const T* locp = loc + sidx * p.L * p.P * 2;
    const T* awp = aw + sidx * p.L * p.P;
    const int P = TP > 0 ? TP : p.P;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < p.L; ++l) {
        const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
        const T* vbase = value + ((long)b * p.S + p.lsi[l]) * ws + (long)m * p.D + dc * 8;
        Corners cs[TP > 0 ? TP : MAX_P];
#pragma unroll
        for (int pt = 0; pt < (TP > 0 ? TP : MAX_P); ++pt)
            if (pt < P) cs[pt] = make_corners(to_f32<T>(locp[(l * P + pt) * 2]), to_f32<T>(locp[(l * P + pt) * 2 + 1]),
                                              to_f32<T>(awp[l * P + pt]), H, W, ws);
        sample_level<T, TP>(vbase, cs, P, acc);
    }
    typename Vec<T>::v8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(acc[e]);
    *(typename Vec<T>::v8*)((T*)p.out + sidx * p.D + dc * 8) = o;
}

// ------------------------------------------------------------------------------------------------ fused
// oa row (b*Q+q): [ offsets M*L*P*2 | ... | logits M*L*P at column oa_logit_off ], ref (B,Q,4) f32 (cx,cy,w,h),
// vr (B,L,2) f32 valid ratios (w,h).  loc = ref_xy*vr + off / P * (ref_wh*vr) * 0.5; weights = softmax_{L*P}(logits).
template <typename T, int TL, int TP>
__global__ __launch_bounds__(256) void msda_fused_kernel(MsdaParams p) {
    int b, q, m, dc;
    if (!msda_work_item(p, b, q, m, dc)) return;
    constexpr int LP = TL * TP;
    const T* value = (const T*)p.value;
    const long ws = (long)p.M * p.D;
    const long row = (long)b * p.Q + q;
    const T* offp = (const T*)p.oa + row * p.ld_oa + (long)m * LP * 2;
    const T* logp = (const T*)p.oa + row * p.ld_oa + p.oa_logit_off + (long)m * LP;
    float lg[LP], mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int i = 0; i < LP; ++i) { lg[i] = to_f32<T>(logp[i]); mx = fmaxf(mx, lg[i]); }
#pragma unroll
    for (int i = 0; i < LP; ++i) { lg[i] = __expf(lg[i] - mx); sum += lg[i]; }
    const float inv = 1.f / sum;
    const float* rf = p.ref + row * 4;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < TL; ++l) {
        const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
        const float vx = p.vr[((long)b * TL + l) * 2], vy = p.vr[((long)b * TL + l) * 2 + 1];
        const float cx = rf[0] * vx, cy = rf[1] * vy, bw = rf[2] * vx, bh = rf[3] * vy;
        const T* vbase = value + ((long)b * p.S + p.lsi[l]) * ws + (long)m * p.D + dc * 8;
        Corners cs[TP];
#pragma unroll
        for (int pt = 0; pt < TP; ++pt) {
            const float ox = to_f32<T>(offp[(l * TP + pt) * 2]), oy = to_f32<T>(offp[(l * TP + pt) * 2 + 1]);
            cs[pt] = make_corners(cx + ox / (float)TP * bw * 0.5f, cy + oy / (float)TP * bh * 0.5f,
                                  lg[l * TP + pt] * inv, H, W, ws);
        }
        sample_level<T, TP>(vbase, cs, TP, acc);
    }
    typename Vec<T>::v8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = from_f32<T>(acc[e]);
    *(typename Vec<T>::v8*)((T*)p.out + row * ws + (long)m * p.D + dc * 8) = o;
}

void fill_grid(MsdaParams& p, dim3& grid) {
    const long per_img = (long)p.Q * p.M * (p.D >> 3);
    p.chunks_per_img = (int)((per_img + 255) / 256);
    p.xcd_remap = (p.B % 8 == 0) ? 1 : 0;
    grid = dim3((unsigned)((long)p.B * p.chunks_per_img));
}

template <typename T> int launch_plain(MsdaParams p, hipStream_t st) {
    const long n = (long)p.B * p.Q * p.M * p.D;
    if (n == 0) return LWDETR_OK;
    const double bytes = ((double)p.B * p.S * p.M * p.D + (double)p.B * p.Q * p.M * p.L * p.P * 3 + (double)n) * sizeof(T);
    if (p.D % 8 == 0 && p.P <= MAX_P && sizeof(T) <= 4) {
        dim3 grid; fill_grid(p, grid);
        ProfScope ps(KID_MSDA, 0.0, bytes, st);
        if (p.P == 2) hipLaunchKernelGGL((msda_vec8_kernel<T, 2>), grid, dim3(256), 0, st, p);
        else if (p.P == 4) hipLaunchKernelGGL((msda_vec8_kernel<T, 4>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((msda_vec8_kernel<T, 0>), grid, dim3(256), 0, st, p);
    } else {
        long blocks = (n + 255) / 256; if (blocks > 65535 * 16) blocks = 65535 * 16;
        ProfScope ps(KID_MSDA_GENERIC, 0.0, bytes, st);
        hipLaunchKernelGGL((msda_generic_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    }
    return lwdetr_check_launch();
}

template <> int launch_plain<double>(MsdaParams p, hipStream_t st) {
    const long n = (long)p.B * p.Q * p.M * p.D;
    if (n == 0) return LWDETR_OK;
    long blocks = (n + 255) / 256; if (blocks > 65535 * 16) blocks = 65535 * 16;
    ProfScope ps(KID_MSDA_GENERIC, 0.0, 0.0, st);
    hipLaunchKernelGGL((msda_generic_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    return lwdetr_check_launch();
}

template <typename T> int launch_fused(MsdaParams p, hipStream_t st) {
    dim3 grid; fill_grid(p, grid);
    const double bytes = ((double)p.B * p.S * p.M * p.D + (double)p.B * p.Q * p.M * p.L * p.P * 3 +
                          (double)p.B * p.Q * p.M * p.D) * sizeof(T);
    ProfScope ps(KID_MSDA_FUSED, 0.0, bytes, st);
    if (p.L == 1 && p.P == 2) hipLaunchKernelGGL((msda_fused_kernel<T, 1, 2>), grid, dim3(256), 0, st, p);
    else if (p.L == 2 && p.P == 4) hipLaunchKernelGGL((msda_fused_kernel<T, 2, 4>), grid, dim3(256), 0, st, p);
    else if (p.L == 1 && p.P == 4) hipLaunchKernelGGL((msda_fused_kernel<T, 1, 4>), grid, dim3(256), 0, st, p);
    else if (p.L == 2 && p.P == 2) hipLaunchKernelGGL((msda_fused_kernel<T, 2, 2>), grid, dim3(256), 0, st, p);
    else return LWDETR_ERR_UNSUPPORTED;
    return lwdetr_check_launch();
}

// --------------------------------------------------------------------------------------------- backward
// col2im (reference ms_deform_im2col_cuda.cuh:87-160 corner gradients, :846-920 loop and in-range test; host contract
// ms_deform_attn_cuda.cu:83-153). A group of GS = min(64, pow2(D)) lanes owns one (b, q, head); lanes stride the channels,
// so the four corner reads and the grad_value atomics of a sampling point are contiguous, the location / weight reads are
// group-uniform, and the channel reduction of d/d(loc) and d/d(weight) - shared-memory trees plus a second pass in the
// reference (:301-845 has seven block-size variants of it) - is log2(GS) wave shuffles. grad_value uses hardware
// floating-point atomics (unsafeAtomicAdd); like the reference's atomicAdd its summation order is not deterministic.
template <typename T>
__global__ __launch_bounds__(256) void msda_backward_kernel(MsdaParams p, const T* __restrict__ grad_out, T* __restrict__ grad_value,
                                                            T* __restrict__ grad_loc, T* __restrict__ grad_aw, int GS) {
    const int lane = threadIdx.x & 63, gl = lane & (GS - 1);
    const int groups_per_block = 256 / GS;
    const long item = (long)blockIdx.x * groups_per_block + (threadIdx.x / GS);      // (b, q, m)
    const long nitems = (long)p.B * p.Q * p.M;
    const bool live = item < nitems;
    const long it = live ? item : nitems - 1;                                          // dead groups shadow the last item
    const T* value = (const T*)p.value; const T* loc = (const T*)p.loc; const T* aw = (const T*)p.aw;
    const int m = (int)(it % p.M);
    const int b = (int)(it / ((long)p.M * p.Q));
    const long ws = (long)p.M * p.D;
    long wp = it * p.L * p.P;
    for (int l = 0; l < p.L; ++l) {
        const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
        const long voff = ((long)b * p.S + p.lsi[l]) * ws + (long)m * p.D;
        for (int pt = 0; pt < p.P; ++pt, ++wp) {
            const T w_im = loc[2 * wp] * W - (T)0.5, h_im = loc[2 * wp + 1] * H - (T)0.5;
            const T a = aw[wp];
            T ga = 0, gx = 0, gy = 0;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
                const int hl = (int)floor(h_im), wl = (int)floor(w_im);
                const T lh = h_im - hl, lw = w_im - wl, hh = 1 - lh, hw = 1 - lw;
                const bool c1 = hl >= 0 && wl >= 0, c2 = hl >= 0 && wl + 1 <= W - 1;
                const bool c3 = hl + 1 <= H - 1 && wl >= 0, c4 = hl + 1 <= H - 1 && wl + 1 <= W - 1;
                const long o1 = voff + ((long)hl * W + wl) * ws, o2 = o1 + ws, o3 = o1 + (long)W * ws, o4 = o3 + ws;
                for (int c = gl; c < p.D; c += GS) {
                    const T tg = grad_out[it * p.D + c];
                    const T tgv = tg * a;
                    T v1 = 0, v2 = 0, v3 = 0, v4 = 0, gh = 0, gw = 0;
                    if (c1) { v1 = value[o1 + c]; gh -= hw * v1; gw -= hh * v1; if (live) unsafeAtomicAdd(grad_value + o1 + c, hh * hw * tgv); }
                    if (c2) { v2 = value[o2 + c]; gh -= lw * v2; gw += hh * v2; if (live) unsafeAtomicAdd(grad_value + o2 + c, hh * lw * tgv); }
                    if (c3) { v3 = value[o3 + c]; gh += hw * v3; gw -= lh * v3; if (live) unsafeAtomicAdd(grad_value + o3 + c, lh * hw * tgv); }
                    if (c4) { v4 = value[o4 + c]; gh += lw * v4; gw += lh * v4; if (live) unsafeAtomicAdd(grad_value + o4 + c, lh * lw * tgv); }
                    ga += tg * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
                    gx += (T)W * gw * tgv;
                    gy += (T)H * gh * tgv;
                }
            }
            for (int o = GS >> 1; o >= 1; o >>= 1) {
                ga += __shfl_xor(ga, o); gx += __shfl_xor(gx, o); gy += __shfl_xor(gy, o);
            }
            if (live && gl == 0) { grad_aw[wp] = ga; grad_loc[2 * wp] = gx; grad_loc[2 * wp + 1] = gy; }
        }
    }
}

template <typename T> int launch_backward(MsdaParams p, const void* grad_out, void* grad_value, void* grad_loc, void* grad_aw,
                                          hipStream_t st) {
    if (hipMemsetAsync(grad_value, 0, (size_t)p.B * p.S * p.M * p.D * sizeof(T), st) != hipSuccess) return LWDETR_ERR_LAUNCH;
    const long nitems = (long)p.B * p.Q * p.M;
    if (nitems == 0) return LWDETR_OK;
    int gs = 1;
    while (gs < p.D && gs < 64) gs <<= 1;
    const long per_block = 256 / gs, blocks = (nitems + per_block - 1) / per_block;
    if (blocks > 0x7fffffffL) return LWDETR_ERR_BAD_ARG;
    ProfScope ps(KID_MSDA_GENERIC, 0.0, 0.0, st);
    hipLaunchKernelGGL((msda_backward_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, p, (const T*)grad_out, (T*)grad_value,
                       (T*)grad_loc, (T*)grad_aw, gs);
    return lwdetr_check_launch();
}

}  // namespace

extern "C" {

// See include/lwdetr_hip.h for the contract.
int lwdetr_msda_forward(const void* value, const int64_t* shapes, const int64_t* level_start, const void* loc,
                        const void* attn, void* out, int B, int S, int M, int D, int L, int Q, int P, int dtype,
                        void* hip_stream) {
    if (B < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Q < 0 || P <= 0) return LWDETR_ERR_BAD_ARG;
    if ((long)B * Q * M * D == 0) return LWDETR_OK;
    if (!value || !shapes || !level_start || !loc || !attn || !out) return LWDETR_ERR_BAD_ARG;
    MsdaParams p = {};
    p.value = value; p.shapes = shapes; p.lsi = level_start; p.loc = loc; p.aw = attn; p.out = out;
    p.B = B; p.S = S; p.M = M; p.D = D; p.L = L; p.Q = Q; p.P = P;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F32: return launch_plain<float>(p, st);
        case DT_F16: return launch_plain<f16>(p, st);
        case DT_BF16: return launch_plain<bf16>(p, st);
        case 3: return launch_plain<double>(p, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

int lwdetr_msda_backward(const void* value, const int64_t* shapes, const int64_t* level_start, const void* loc,
                         const void* attn, const void* grad_out, void* grad_value, void* grad_loc, void* grad_attn,
                         int B, int S, int M, int D, int L, int Q, int P, int dtype, void* hip_stream) {
    if (B < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Q < 0 || P <= 0) return LWDETR_ERR_BAD_ARG;
    if (dtype != DT_F32 && dtype != 3) return LWDETR_ERR_UNSUPPORTED;          // the reference differentiates float / double only
    if ((long)B * S * M * D == 0 && (long)B * Q * M == 0) return LWDETR_OK;
    if (!value || !shapes || !level_start || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn)
        return LWDETR_ERR_BAD_ARG;
    MsdaParams p = {};
    p.value = value; p.shapes = shapes; p.lsi = level_start; p.loc = loc; p.aw = attn;
    p.B = B; p.S = S; p.M = M; p.D = D; p.L = L; p.Q = Q; p.P = P;
    hipStream_t st = (hipStream_t)hip_stream;
    return dtype == DT_F32 ? launch_backward<float>(p, grad_out, grad_value, grad_loc, grad_attn, st)
                           : launch_backward<double>(p, grad_out, grad_value, grad_loc, grad_attn, st);
}

int lwdetr_msda_fused_forward(const void* value, const int64_t* shapes, const int64_t* level_start, const void* oa,
                              long ld_oa, int logit_col, const float* ref_boxes, const float* valid_ratios, void* out,
                              int B, int S, int M, int D, int L, int Q, int P, int dtype, void* hip_stream) {
    if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || D % 8 != 0 || L <= 0 || Q <= 0 || P <= 0) return LWDETR_ERR_BAD_ARG;
    if (!value || !shapes || !level_start || !oa || !ref_boxes || !valid_ratios || !out) return LWDETR_ERR_BAD_ARG;
    MsdaParams p = {};
    p.value = value; p.shapes = shapes; p.lsi = level_start; p.out = out; p.oa = oa; p.ld_oa = ld_oa;
    p.oa_logit_off = logit_col; p.ref = ref_boxes; p.vr = valid_ratios;
    p.B = B; p.S = S; p.M = M; p.D = D; p.L = L; p.Q = Q; p.P = P;
    hipStream_t st = (hipStream_t)hip_stream;
    switch (dtype) {
        case DT_F32: return launch_fused<float>(p, st);
        case DT_F16: return launch_fused<f16>(p, st);
        case DT_BF16: return launch_fused<bf16>(p, st);
        default: return LWDETR_ERR_UNSUPPORTED;
    }
}

}  // extern "C"
