/* TEST INFRASTRUCTURE ONLY - scalar C restatement of the reference's deformable-attention FORWARD and BACKWARD kernels.
 *
 * Follows the arithmetic of /root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:
 *   bilinear sample with zero padding ............ :33-84  (ms_deform_attn_im2col_bilinear)
 *   per-output loop over levels and points ....... :237-299 (ms_deformable_im2col_gpu_kernel)
 * and the host contract of ms_deform_attn_cuda.cu:20-80 (value (B,S,M,D), loc (B,Q,M,L,P,2) in (x,y)
 * order, weights (B,Q,M,L,P), output (B,Q,M*D)). The reference ships no CPU build of this op
 * (src/cpu/ms_deform_attn_cpu.cpp:16-35 only throws), so this file is pinned against the reference's own
 * PyTorch core through tests/golden/msda_op_kat.npz (tests/test_msda_oracle.py).
 * Backward (col2im): :87-160 (ms_deform_attn_col2im_bilinear: corner gradients, d/d(loc), d/d(weight)) and :846-920
 * (ms_deformable_col2im_gpu_kernel_gm: loop order and in-range test), host contract ms_deform_attn_cuda.cu:83-153;
 * the channel reduction of grad_sampling_loc / grad_attn_weight is a plain sum here. Pinned against autograd through the
 * reference's PyTorch core (same golden file).
 * Instantiated for double and float; used by tests and by bench.py's cpu_baseline leg only.
 */
#include <math.h>
#include <stdint.h>

#define DEFINE_MSDA(NAME, T)                                                                         \
static T NAME##_bilinear(const T *v, int H, int W, int M, int D, T h, T w, int m, int c) {          \
    const int h_low = (int)floor((double)h), w_low = (int)floor((double)w);                          \
    const int h_high = h_low + 1, w_high = w_low + 1;                                                \
    const T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                                \
    const int64_t ws = (int64_t)M * D, hs = (int64_t)W * ws, base = (int64_t)m * D + c;              \
    T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                                \
    if (h_low >= 0 && w_low >= 0) v1 = v[h_low * hs + w_low * ws + base];                            \
    if (h_low >= 0 && w_high <= W - 1) v2 = v[h_low * hs + w_high * ws + base];                      \
    if (h_high <= H - 1 && w_low >= 0) v3 = v[h_high * hs + w_low * ws + base];                      \
    if (h_high <= H - 1 && w_high <= W - 1) v4 = v[h_high * hs + w_high * ws + base];                \
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;                                \
}                                                                                                    \
int NAME(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *aw,       \
         T *out, int B, int S, int M, int D, int L, int Q, int P) {                                  \
    for (int b = 0; b < B; ++b)                                                                      \
      for (int q = 0; q < Q; ++q)                                                                    \
        for (int m = 0; m < M; ++m)                                                                  \
          for (int c = 0; c < D; ++c) {                                                              \
            int64_t wp = (((int64_t)b * Q + q) * M + m) * L * P, lp = wp * 2;                        \
            T col = 0;                                                                               \
            for (int l = 0; l < L; ++l) {                                                            \
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                        \
                const T *v = value + ((int64_t)b * S + lsi[l]) * M * D;                              \
                for (int p = 0; p < P; ++p, ++wp, lp += 2) {                                         \
                    const T h_im = loc[lp + 1] * H - (T)0.5, w_im = loc[lp] * W - (T)0.5;            \
                    if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                              \
                        col += NAME##_bilinear(v, H, W, M, D, h_im, w_im, m, c) * aw[wp];            \
                }                                                                                    \
            }                                                                                        \
            out[(((int64_t)b * Q + q) * M + m) * D + c] = col;                                       \
          }                                                                                          \
    return 0;                                                                                        \
}

DEFINE_MSDA(msda_ref_f64, double)
DEFINE_MSDA(msda_ref_f32, float)

#define DEFINE_MSDA_BWD(NAME, T)                                                                     \
int NAME(const T *value, const int64_t *shapes, const int64_t *lsi, const T *loc, const T *aw,       \
         const T *grad_out, T *grad_value, T *grad_loc, T *grad_aw,                                  \
         int B, int S, int M, int D, int L, int Q, int P) {                                          \
    for (int64_t i = 0; i < (int64_t)B * S * M * D; ++i) grad_value[i] = 0;                          \
    for (int64_t i = 0; i < (int64_t)B * Q * M * L * P; ++i) { grad_aw[i] = 0; grad_loc[2 * i] = 0; grad_loc[2 * i + 1] = 0; } \
    for (int b = 0; b < B; ++b)                                                                      \
      for (int q = 0; q < Q; ++q)                                                                    \
        for (int m = 0; m < M; ++m)                                                                  \
          for (int c = 0; c < D; ++c) {                                                              \
            const T top_grad = grad_out[(((int64_t)b * Q + q) * M + m) * D + c];                     \
            int64_t wp = (((int64_t)b * Q + q) * M + m) * L * P;                                     \
            for (int l = 0; l < L; ++l) {                                                            \
                const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                        \
                const int64_t voff = ((int64_t)b * S + lsi[l]) * M * D;                              \
                const T *v = value + voff; T *gv = grad_value + voff;                                \
                for (int p = 0; p < P; ++p, ++wp) {                                                  \
                    const T h = loc[2 * wp + 1] * H - (T)0.5, w = loc[2 * wp] * W - (T)0.5;          \
                    if (!(h > -1 && w > -1 && h < H && w < W)) continue;                             \
                    const int h_low = (int)floor((double)h), w_low = (int)floor((double)w);          \
                    const int h_high = h_low + 1, w_high = w_low + 1;                                \
                    const T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                \
                    const int64_t ws = (int64_t)M * D, hs = (int64_t)W * ws, base = (int64_t)m * D + c; \
                    const T tgv = top_grad * aw[wp];                                                 \
                    T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                \
                    if (h_low >= 0 && w_low >= 0) { const int64_t o = h_low * hs + w_low * ws + base; v1 = v[o]; gh -= hw * v1; gw -= hh * v1; gv[o] += hh * hw * tgv; } \
                    if (h_low >= 0 && w_high <= W - 1) { const int64_t o = h_low * hs + w_high * ws + base; v2 = v[o]; gh -= lw * v2; gw += hh * v2; gv[o] += hh * lw * tgv; } \
                    if (h_high <= H - 1 && w_low >= 0) { const int64_t o = h_high * hs + w_low * ws + base; v3 = v[o]; gh += hw * v3; gw -= lh * v3; gv[o] += lh * hw * tgv; } \
                    if (h_high <= H - 1 && w_high <= W - 1) { const int64_t o = h_high * hs + w_high * ws + base; v4 = v[o]; gh += lw * v4; gw += lh * v4; gv[o] += lh * lw * tgv; } \
                    grad_aw[wp] += top_grad * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4); \
                    grad_loc[2 * wp] += W * gw * tgv;                                                \
                    grad_loc[2 * wp + 1] += H * gh * tgv;                                            \
                }                                                                                    \
            }                                                                                        \
          }                                                                                          \
    return 0;                                                                                        \
}

DEFINE_MSDA_BWD(msda_ref_bwd_f64, double)
DEFINE_MSDA_BWD(msda_ref_bwd_f32, float)
