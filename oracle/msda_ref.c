/* TEST INFRASTRUCTURE ONLY - scalar C restatement of the reference's deformable-attention FORWARD kernel.
 *
 * Follows the arithmetic of /root/reference/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:
 *   bilinear sample with zero padding ............ :33-84  (ms_deform_attn_im2col_bilinear)
 *   per-output loop over levels and points ....... :237-299 (ms_deformable_im2col_gpu_kernel)
 * and the host contract of ms_deform_attn_cuda.cu:20-80 (value (B,S,M,D), loc (B,Q,M,L,P,2) in (x,y)
 * order, weights (B,Q,M,L,P), output (B,Q,M*D)). The reference ships no CPU build of this op
 * (src/cpu/ms_deform_attn_cpu.cpp:16-35 only throws), so this file is pinned against the reference's own
 * PyTorch core through tests/golden/msda_op_kat.npz (tests/test_msda_oracle.py).
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
