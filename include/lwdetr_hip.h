/* lwdetr_hip.h - C ABI of liblwdetr_hip.so (hand-written gfx950 kernels for the LW-DETR forward path).
 *
 * Conventions for every entry point:
 *   - all tensor pointers are DEVICE pointers, borrowed for the duration of the call (no ownership transfer);
 *   - outputs are caller-allocated and fully overwritten; nothing is allocated, nothing synchronises the host;
 *   - work is enqueued on `hip_stream` (a hipStream_t passed as void*, NULL = default stream); stateless and
 *     re-entrant per stream;
 *   - return value: 0 on success, negative LWDETR_ERR_* otherwise (launch failures are reported from
 *     hipGetLastError(), never just printed - the reference only printf's them, ms_deform_im2col_cuda.cuh:948-952);
 *   - dtype codes: 0 = float32, 1 = float16, 2 = bfloat16 (3 = float64, deformable-attention op only).
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   lwdetr_msda_forward        models/ops/src/ms_deform_attn.h:19-35 (ms_deform_attn_forward, pybind
 *                              models/ops/src/vision.cpp:13-16) -> cuda/ms_deform_attn_cuda.cu:20-80 ->
 *                              cuda/ms_deform_im2col_cuda.cuh:237-299
 *   lwdetr_msda_backward       models/ops/src/ms_deform_attn.h:37-60 (ms_deform_attn_backward) ->
 *                              cuda/ms_deform_attn_cuda.cu:83-153 -> cuda/ms_deform_im2col_cuda.cuh:87-160, :846-920
 *   lwdetr_msda_fused_forward  models/ops/modules/ms_deform_attn.py:117-142 (softmax, location arithmetic, op call)
 *   lwdetr_gemm                torch.nn.functional.linear / conv2d / conv_transpose2d call sites of
 *                              models/backbone/vit.py:79-83,:123-138, timm Mlp, models/backbone/projector.py:85-132,
 *                              :177-193, models/transformer.py:28-39,:231-240, models/attention.py:507-560,
 *                              models/lwdetr.py:149-159 - with the bias / activation / LayerScale / residual /
 *                              layout epilogues fused
 *   lwdetr_attention           models/backbone/vit.py:130-137 (window and global softmax(QK^T)V) and
 *                              models/attention.py:563-606 (decoder self-attention)
 *   lwdetr_mlp_fused           models/backbone/vit.py:217-218 (+ timm.models.layers.Mlp: fc1 -> GELU -> fc2)
 *   lwdetr_vit_block           models/backbone/vit.py:138, :199-218, :123-130 (projection + MLP + next block's norm1 / QKV)
 *   lwdetr_ffn_partial/_finish models/transformer.py:507-512, :397-400 (decoder FFN + norm3 + decoder.norm)
 *   lwdetr_layernorm           nn.LayerNorm call sites (vit.py:199,:217; transformer.py:231,:499,:511,:516,:398) and
 *                              the channel LayerNorm of models/backbone/projector.py:21-47
 */
#ifndef LWDETR_HIP_H
#define LWDETR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LWDETR_OK 0
#define LWDETR_ERR_BAD_ARG (-1)
#define LWDETR_ERR_UNSUPPORTED (-2)
#define LWDETR_ERR_LAUNCH (-3)

/* out[b,q,m*D+c] = sum_l sum_p attn[b,q,m,l,p] * bilinear(value[b, lvl l, :, m, c], (loc_x*W_l-.5, loc_y*H_l-.5)), zero pad.
 * value (B,S,M,D); shapes (L,2) int64 (H,W); level_start (L) int64; loc (B,Q,M,L,P,2) (x,y); attn (B,Q,M,L,P); out (B,Q,M*D). */
int lwdetr_msda_forward(const void* value, const int64_t* shapes, const int64_t* level_start, const void* loc,
                        const void* attn, void* out, int B, int S, int M, int D, int L, int Q, int P, int dtype,
                        void* hip_stream);

/* Backward of the op (reference ms_deform_attn_backward, models/ops/src/ms_deform_attn.h:37-60 ->
 * cuda/ms_deform_attn_cuda.cu:83-153 -> cuda/ms_deform_im2col_cuda.cuh:87-160, :846-920): grad_out (B,Q,M*D) ->
 * grad_value (B,S,M,D) (zeroed here, then accumulated with atomics), grad_loc (B,Q,M,L,P,2), grad_attn (B,Q,M,L,P),
 * all fully written. dtype 0 (float32) or 3 (float64), as in the reference. */
int lwdetr_msda_backward(const void* value, const int64_t* shapes, const int64_t* level_start, const void* loc,
                         const void* attn, const void* grad_out, void* grad_value, void* grad_loc, void* grad_attn,
                         int B, int S, int M, int D, int L, int Q, int P, int dtype, void* hip_stream);

/* Model-path variant: oa is the (B*Q, ld_oa) output of the fused sampling_offsets|attention_weights Linear
 * (offsets at column 0: M*L*P*2 values, logits at column logit_col: M*L*P values); ref_boxes (B,Q,4) f32 (cx,cy,w,h);
 * valid_ratios (B,L,2) f32 (w,h). Computes softmax over L*P and loc = ref_xy + off / P * ref_wh * 0.5 in-kernel. D % 8 == 0. */
int lwdetr_msda_fused_forward(const void* value, const int64_t* shapes, const int64_t* level_start, const void* oa,
                              long ld_oa, int logit_col, const float* ref_boxes, const float* valid_ratios, void* out,
                              int B, int S, int M, int D, int L, int Q, int P, int dtype, void* hip_stream);

/* ---- fused MFMA GEMM: out = epilogue(A_view(M,K) * W(N,K)^T), see lwdetr_gemm_desc ---------------------------- */
typedef struct {
    int winmajor;      /* 0: rows are raster (b,y,x) tokens; 1: window-major padded rows (see DESIGN.md "data layout") */
    int Hp, Wp, Twp;
} lwdetr_tok_layout;

enum { LWDETR_A_PLAIN = 0, LWDETR_A_CONV3x3 = 1, LWDETR_A_PATCH16 = 2 };
enum { LWDETR_ACT_NONE = 0, LWDETR_ACT_RELU = 1, LWDETR_ACT_GELU = 2, LWDETR_ACT_SILU = 3 };
enum { LWDETR_OUT_LINEAR = 0, LWDETR_OUT_HEADS = 1, LWDETR_OUT_HEADS_T = 2, LWDETR_OUT_TOKMAP = 3,
       LWDETR_OUT_DECONV2x2 = 4 };

typedef struct {           /* one column segment [n_begin, n_end) of the output */
    void* out;             /* destination base */
    void* out2;            /* optional second LINEAR destination (ViT feature taps), row stride ld2 */
    const void* res;       /* optional residual, same dtype as out, addressed LINEAR with ldres; row = m % res_mod if res_mod>0 */
    const float* bias;     /* optional f32, indexed by n - n_begin; 16-byte aligned and readable up to ceil8(n_end-n_begin) */
    const float* gamma;    /* optional f32 LayerScale (same padding rule): out = res + gamma * act(acc + bias) * scale */
    const uint8_t* rowmask;/* optional (M): rows with mask == 0 get acc = 0 before bias (rowmask_after = 0, the reference's
                              masked_fill of the INPUT row) or a zero OUTPUT row (rowmask_after = 1, masked_fill of the result) */
    float scale;           /* multiplies (acc + bias) */
    int act;
    int mode;              /* LWDETR_OUT_* */
    int n_begin, n_end;
    long ldo, ld2, ldres;
    int res_mod;
    int rowmask_after;
    int p0, p1, p2;        /* HEADS/HEADS_T: p0 = tokens per image Tp, p1 = head_dim, p2 = heads */
    lwdetr_tok_layout in_tok, out_tok;   /* TOKMAP / DECONV2x2: row decode / encode layouts */
    long out_batch_stride; /* TOKMAP / DECONV2x2: elements between images in out (0 = dense) */
    long out_row_offset;   /* TOKMAP / DECONV2x2: first row inside an image (level offset into `memory`) */
    /* LayerNorm folded into the GEMM (round 5; `nn.LayerNorm` in front of a Linear, models/backbone/vit.py:199, :217): with W' = W diag(g),
       b' = b + W beta packed by the host, LN(x) W^T + b = rstd_m (x_m . W'_n - mean_m colsum_n) + b'_n - the GEMM reads the RAW rows and the
       epilogue applies the row statistics: acc <- (acc - mean * ln_colsum[n]) * rstd before bias / activation. All segments of a launch or
       none; served by the 256 x 256 large-tile kernel only (16-bit, PLAIN A, K % 64 == 0, segment boundaries at multiples of 256):
       LWDETR_ERR_UNSUPPORTED for any other shape. */
    const float* ln_stats; /* optional (2, M) f32, planar: ln_stats[m] = mean, ln_stats[M + m] = rstd of A row m (lwdetr_row_stats); NULL = plain GEMM */
    const float* ln_colsum;/* with ln_stats: f32 sum over k of W[n][k] as stored (16-bit rounded), indexed by n - n_begin, padded like bias */
} lwdetr_gemm_seg;

typedef struct {
    const void* A; const void* A2;   /* A2 optional: A_view = A + A2 (same shape, PLAIN only) */
    const void* W;                   /* (N, K) row-major, same dtype as A */
    int M, N, K;
    long lda;
    int a_mode;
    /* CONV3x3: A is (B, Hin, Win, ldc) tokens in layout a_tok with Cin channels at column a_col0; stride 1|2; K = 9*Cin.
       PATCH16: A is the (B,3,H,W) image, rows are window-major tokens of a_tok; K = 768. */
    lwdetr_tok_layout a_tok;
    int conv_cin, conv_stride, a_col0, conv_hout, conv_wout;
    int img_h, img_w;
    int nseg;
    lwdetr_gemm_seg seg[3];
    /* Split-K for few-row GEMMs with a long contraction (round 5): splitk >= 2 asks for that many workgroups per 64 x 64 tile, each over a
       contiguous range of k-stages; honoured only where the launch takes the 64 x 64 DMA ring kernel (16-bit, no A2), ignored elsewhere.
       splitk_ws: 16-byte aligned workspace of tiles * splitk * 4352 floats (f32 partial tiles) followed by `tiles` ints (arrival counters),
       tiles = ceil(M / 64) * ceil(N / 64); the counters must be ZERO before the first launch (every launch leaves them zero), the workspace
       belongs to one launch at a time. Results do not depend on the arrival order (slabs are summed in slice order). */
    void* splitk_ws;
    int splitk;
} lwdetr_gemm_desc;

int lwdetr_gemm(const lwdetr_gemm_desc* desc, int dtype, void* hip_stream);
/* Few-row form (round 6; the single-image latency path): the same descriptor as lwdetr_gemm with W FRAGMENT-MAJOR - [N / 16][K / 32][16][32],
 * lwdetr_amd.kernels.pack_frag16 of the (N, K) matrix lwdetr_gemm takes. A workgroup owns 16 rows x 128 columns, a wave loads the MFMA fragments of a
 * third of the contraction straight from L2 before it multiplies them: no LDS ring, no barriers - the six 3x3 convolutions of the projector
 * (models/backbone/projector.py:101-132) at one 640 x 640 image 19.3 -> ~9 us each. Serves: 16-bit, M <= 8192, a_mode PLAIN or CONV3x3 (raster rows,
 * stride 1 | 2, Cin in {128, 192}), K % 32 == 0, N % 16 == 0, ONE LINEAR segment (bias, activation, scale, gamma, residual, second destination;
 * no row mask); LWDETR_ERR_UNSUPPORTED otherwise. Same arithmetic as lwdetr_gemm up to the f32 summation order of the contraction. */
int lwdetr_gemm_few(const lwdetr_gemm_desc* desc, int dtype, void* hip_stream);
/* Row statistics of x (M, C) for the LayerNorm-folded GEMM: stats[m] = mean, stats[M + m] = 1 / sqrt(var + eps) (planar - interleaved pairs made
 * hipcc broadcast the high half of a register pair into packed-f32 epilogue arithmetic, the instruction form tools/check_isa.py refuses),
 * two-pass f32 on the stored values -
 * the arithmetic of lwdetr_layernorm without its output pass (half its HBM traffic). C % 8 == 0 (16-bit) / C % 4 == 0 (f32). */
int lwdetr_row_stats(const void* x, long ldx, long M, int C, float eps, float* stats, int dtype, void* hip_stream);
/* Kernel selection override for tests / tuning (process-wide): big_mode -1 = default (environment LWDETR_GEMM_BIG, else
 * shape thresholds), 0 = never use the 256-row large-tile kernel, 2 = use it whenever the shape is legal for it,
 * 32 / 64 = as 2 with that stage depth. Results do not depend on it beyond f32 summation order. */
void lwdetr_gemm_tuning(int big_mode);
/* 1 if the library was built with -DLWDETR_EXPERIMENTS: the kernel forms that were built, validated and measured SLOWER than what the launch plan
 * uses (round 5: the 4-wave / 128-row large-tile GEMM, the LayerNorm-folded GEMM epilogue behind ln_stats, split-K behind splitk, the LDS window-tile
 * attention kernel) are only compiled then. The default library (0) answers such requests with LWDETR_ERR_UNSUPPORTED or takes the ordinary kernel. */
int lwdetr_has_experiments(void);
/* The persistent large-tile kernel (round 6, csrc/gemm_pt.hip: a workgroup walks its 256 x 256 tiles, the DMA ring runs on across tile boundaries,
 * the epilogue goes straight from the accumulators to memory) serves the 16-bit plain-A launches whose segments are whole 256-column tiles of
 * LINEAR / HEADS / HEADS_T outputs - the four Linear layers of a C = 768 ViT block (models/backbone/vit.py:123-138, :217-218). Process-wide
 * override for tests / tuning: pt_mode -1 = default (environment LWDETR_GEMM_PT, read once; else 1), 0 = never, 1 = at the sizes where the
 * 256-row tile pays, 2 = whenever the shape is legal; tuning: + 256 * w sets the start-skew window to w ticks of 10 ns. Results equal the
 * large-tile kernel's up to f32 rounding of the epilogue (bias first, one scale multiplication). */
void lwdetr_gemm_pt_tuning(int pt_mode);
/* launches of the persistent kernel by this process so far (tests assert which kernel served a shape) */
long lwdetr_gemm_pt_count(void);

/* ---- fused softmax(QK^T)V, flash-style, MFMA ------------------------------------------------------------------- */
typedef struct {
    const void* Q;      /* (B, heads, Tp, hd)  pre-scaled by hd^-0.5 * log2(e) */
    const void* K;      /* (B, heads, Tp, hd) */
    const void* VT;     /* (B, heads, hd, Tp) */
    void* out;          /* (B*Tp, ldo) row-major, head h writes columns [h*hd, (h+1)*hd) */
    long ldo;
    int B, heads, hd, Tp;
    int seqs_per_img;   /* 16 (window attention) or 1 */
    int seq_tok_stride; /* tokens between consecutive sequences of one image */
    int keys_per_seq;   /* tokens (incl. pad rows) spanned by one sequence; >= 8 and a multiple of 4 */
    int sub_stride, sub_len; /* key j is real iff (j % sub_stride) < sub_len */
    int kind;           /* 0 window, 1 global, 2 decoder self-attention (profiling label only) */
    int vt_slack;       /* nonzero: at least 8 readable bytes follow the VT tensor (lets the LDS-ring kernel serve sequences
                           whose length is not a multiple of 8: it fetches V^T in whole 16-byte runs) */
} lwdetr_attn_desc;

int lwdetr_attention(const lwdetr_attn_desc* desc, int dtype, void* hip_stream);
/* Kernel selection override for tests / tuning (process-wide): lds_mode -1 = default (environment LWDETR_ATTN_LDS, else 2),
 * 0 = never use the LDS-ring kernel, 1 = sequences of >= 512 keys, 2 = also >= 192-key windows, 3 = everything >= 64 keys. */
void lwdetr_attention_tuning(int lds_mode);
/* LDS-ring kernel shape override for tests / tuning (process-wide): 0 = default (environment LWDETR_ATTN_LDS_CFG, else by head
 * dimension), otherwise 1000 (U - 1) + 100 QT + NW = 32 QT queries per wave, NW waves per workgroup, U 64-key blocks per ring step;
 * a shape that is not compiled in (or does not fit LDS at this head dimension) makes lwdetr_attention return LWDETR_ERR_UNSUPPORTED. */
void lwdetr_attention_tuning_cfg(int cfg);

/* ---- row LayerNorm: out[r,:] = (x[r,:]-mean)/sqrt(var+eps)*gamma+beta; biased variance; C % 4 == 0 ------------- */
/* rows_per_batch / out_batch_stride / out_row_offset let the projector write straight into `memory` (B,S,d). */
int lwdetr_layernorm(const void* x, long ldx, const float* gamma, const float* beta, void* out, long ldo, long M,
                     int C, float eps, long rows_per_batch, long out_batch_rows, long out_row_offset, int dtype,
                     void* hip_stream);

/* out1 = LN1(x), out2 = LN2(out1) in one pass over the row; bit-identical to two lwdetr_layernorm launches (the second reading
 * out1). Decoder layer tail: norm3 followed by the shared decoder.norm (models/transformer.py:397-400, :466-517). */
int lwdetr_layernorm_chain(const void* x, long ldx, const float* gamma1, const float* beta1, float eps1, void* out1, long ldo1,
                           const float* gamma2, const float* beta2, float eps2, void* out2, long ldo2, long M, int C,
                           int dtype, void* hip_stream);

/* ---- decoder FFN in two launches, hidden activation on chip --------------------------------------------------------
 * Replaces models/transformer.py:507-512 (linear2(dropout(relu(linear1(tgt)))) + residual + norm3) and the shared
 * decoder.norm that follows (:397-400). lwdetr_ffn_partial: the grid is (token tiles) x (splits of the hidden dimension);
 * every workgroup keeps ReLU(x W1^T + b1) of its hidden slice in registers and writes the f32 partial product with W2 to
 * partial[split] (M, C) - the 300-queries-per-image row count alone cannot fill 256 CUs, the hidden split can.
 * lwdetr_ffn_finish: x + b2 + sum of the partial slabs (rounded to the 16-bit dtype, as the unfused GEMM epilogue rounds),
 * out1 = LN1(.), optionally out2 = LN2(out1). w1 (hid, C) plain row-major; w2_chunked (hid/32, C, 32) as for
 * lwdetr_mlp_fused. C in {256, 384}, hid % 64 == 0, 16-bit dtypes. lwdetr_ffn_splits returns the number of slabs the
 * launch for (M, C, hid) writes (> 0), or -error. lwdetr_ffn_finish: out1 MAY alias x (every row is read completely before it is
 * written; the engine updates the decoder stream in place); out2 must not alias x or out1. */
int lwdetr_ffn_splits(long M, int C, int hid, int dtype);
int lwdetr_ffn_partial(const void* x, long ldx, const void* w1, const float* b1, const void* w2_chunked, float* partial, long M,
                       int C, int hid, int dtype, void* hip_stream);
int lwdetr_ffn_finish(const void* x, long ldx, const float* partial, int splits, const float* b2, const float* gamma1,
                      const float* beta1, float eps1, void* out1, long ldo1, const float* gamma2, const float* beta2, float eps2,
                      void* out2, long ldo2, long M, int C, int dtype, void* hip_stream);

/* ---- fused ViT MLP: x <- x + gamma2 * fc2(GELU(fc1(LN(x)))), one launch, hidden activation stays on chip ---------
 * Replaces models/backbone/vit.py:217-218 (norm2 -> timm Mlp -> gamma_2 -> residual). Weight packing (host, once):
 *   w1_folded (4C, C) = fc1.weight * norm2.weight[None, :],  b1_folded = fc1.bias + fc1.weight @ norm2.bias (f32),
 *   w2_chunked (4C/32, C, 32) = fc2.weight.view(C, 4C/32, 32)[:, :, perm].permute(1, 0, 2): each 32-wide hidden chunk
 *   contiguous, inside a chunk perm = [4g+e+16*hi for g in 0..3 for hi in 0..1 for e in 0..3] (the kernel's MFMA k-slots).
 * x is updated in place; out2 (optional, row stride ld2) receives a copy (ViT feature taps); stats_out (optional, (M,2)
 * f32) receives mean and 1/sqrt(var+eps_next) of the updated rows for the next block's LayerNorm. C in {192, 384}.
 * Optional fused attention output projection (vit.py:138, :206-216): when att != NULL the kernel first computes
 * x <- x + gamma1 * (att @ wp^T + bp) (att (M,C) row stride ldatt, wp (C,C) = attn.proj.weight, bp / gamma1 f32 (C));
 * w1_folded must then have its columns permuted inside every 32-chunk with the same `perm` as w2_chunked.
 * Optional chained LayerNorm + QKV of the NEXT block (vit.py:199, :123-130) on the updated rows (needs att != NULL):
 * wqkv_next (3C, C) = next qkv.weight * next norm1.weight[None,:] with the same per-32-chunk column permutation,
 * bqkv_next (3C) f32 = [q_bias, 0, v_bias] + qkv.weight @ norm1.bias; writes Q (pre-scaled by qscale), K as
 * (B, heads, Tp, hd) and V^T as (B, heads, hd, Tp) exactly like lwdetr_gemm's HEADS / HEADS_T epilogues (LN eps = eps_next). */
int lwdetr_mlp_fused(void* x, long ldx, const void* w1_folded, const float* b1_folded, const void* w2_chunked,
                     const float* b2, const float* gamma2, void* out2, long ld2, float* stats_out, long M, int C,
                     float eps, float eps_next, const void* att, long ldatt, const void* wp, const float* bp,
                     const float* gamma1, const void* wqkv_next, const float* bqkv_next, void* q_out, void* k_out,
                     void* vt_out, float qscale, int heads, int hd, int Tp, int dtype, void* hip_stream);

/* The few-token form of lwdetr_mlp_fused with the attention projection (round 6; 16-bit, C = 192, M < 12800: the single-image latency path) on
 * FRAGMENT-MAJOR weights: w1_frag, wp_frag, wqkv_frag_next = the w1_folded / wp / wqkv_next of lwdetr_mlp_fused re-laid out as
 * [R / 16][C / 32][16][32] (lwdetr_amd.kernels.pack_frag16), so that each 16 x 32 MFMA fragment the kernel loads straight from L2 is one
 * contiguous KB instead of 16 half lines. Everything else as lwdetr_mlp_fused (att required); results are bit-identical to it.
 * LWDETR_ERR_UNSUPPORTED outside (C, dtype, M) above. */
int lwdetr_vit_block_few(void* x, long ldx, const void* w1_frag, const float* b1_folded, const void* w2_chunked, const float* b2,
                         const float* gamma2, void* out2, long ld2, float* stats_out, long M, int C, float eps, float eps_next,
                         const void* att, long ldatt, const void* wp_frag, const float* bp, const float* gamma1,
                         const void* wqkv_frag_next, const float* bqkv_next, void* q_out, void* k_out, void* vt_out, float qscale,
                         int heads, int hd, int Tp, int dtype, void* hip_stream);

/* ---- fused ViT block tail (round 3; 16-bit dtypes, C in {192, 384}, M % 4 == 0) --------------------------------------
 * Replaces, per ViT block, models/backbone/vit.py:138 + :206-216 (attention output projection, gamma_1, residual),
 * :217-218 (norm2 -> timm Mlp -> gamma_2 -> residual) and, with has_qkv, :199 + :123-130 of the NEXT block (norm1, QKV with
 * q_bias / v_bias; Q pre-scaled by qscale, Q / K as (B, heads, Tp, hd), V^T as (B, heads, hd, Tp) like lwdetr_gemm's HEADS /
 * HEADS_T epilogues). x (M, C) is updated in place, att (M, C) is the attention output; out2 (optional) receives a copy
 * of the new rows (feature taps), stats_out (optional, (M, 2) f32) mean and 1/sqrt(var + eps_next) of the new rows.
 * wstream / vec: lwdetr_amd.kernels.pack_vit_block (all weights of the block as one stream of 1 KB MFMA fragments in
 * consumption order; f32 vectors b1' | bp | g1 | 1/g1 | b2 | 1/g2 | g2 | bqkv'); sizes from the two helpers below.
 * hd must be a power of two; gamma_1 / gamma_2 must be non-zero (the kernel divides by them in f32). */
long lwdetr_vit_block_stream_bytes(int C, int has_qkv);
long lwdetr_vit_block_vec_floats(int C);
int lwdetr_vit_block(void* x, long ldx, const void* att, long ldatt, const void* wstream, const float* vec, void* out2,
                     long ld2, float* stats_out, long M, int C, float eps, float eps_next, int has_qkv, void* q_out,
                     void* k_out, void* vt_out, float qscale, int heads, int hd, int Tp, int dtype, void* hip_stream);

/* ---- row-local chains of Linear stages in one launch (round 4; 16-bit dtypes) ------------------------------------------
 * lwdetr_enc_chain: everything the forward does to ALL S encoder tokens between the projector's C2f block and the two-stage
 * top-k, one launch: [k5 != 0: C2f.cv2 1x1 convolution (BatchNorm folded) + SiLU + channel LayerNorm -> `memory`
 * (models/backbone/projector.py:117-132, :21-47)] -> value_proj of all `nl` decoder layers with the padding mask on the OUTPUT
 * rows (models/ops/modules/ms_deform_attn.py:110-114) -> enc_output Linear on the rows with invalid proposals zeroed +
 * enc_output_norm -> output_memory `om` (models/transformer.py:113-116, :231) -> enc_out_class_embed -> class logits `cls`
 * (row stride ld_cls >= 96, columns >= ncls are zero weight pads) and their row maximum `cls_max` (f32; :244-246).
 * in: k5 != 0: (M, ld_in) rows holding the k5 = 5 * d / 2 channels of the C2f concat of one level, image b = row / npix,
 * destination row b * S + lsi + row % npix of the (total_rows, D) tensors; k5 == 0: `memory` rows themselves (npix = S, lsi = 0).
 * wstream / vec: lwdetr_amd.kernels.pack_enc_chain (4 KB pieces of 32 output channels x 64 k-slots in MFMA lane order, consumption
 * order cv2 | values | enc_output | class, + 2 zero pieces; f32 vectors [b2 | ln_w | ln_b] | b_enc | g_enc | be_enc | b_cls[96] | b_val);
 * sizes from the helpers. D in {256, 384}; k5 in {0, 640 (D = 256)}. */
long lwdetr_enc_chain_vec_floats(int D, int k5);
long lwdetr_enc_chain_pieces(int D, int k5, int nl);
int lwdetr_enc_chain(const void* in, long ld_in, int k5, void* memory, void* om, void* cls, long ld_cls, float* cls_max,
                     void* const* values, int nl, const unsigned char* rowvalid, const unsigned char* notpad,
                     const void* wstream, const float* vec, long M, int D, int npix, int S, int lsi, long total_rows,
                     int ncls, float eps_p, float eps_e, int dtype, void* hip_stream);

/* lwdetr_row_chain: a run-time program of up to 6 Linear stages over rows that never mix (16-bit dtypes, D in {256, 384}); replaces
 * the decoder's small GEMM + LayerNorm launches: self_attn.out_proj + residual + norm1 (+ `tgt + query_pos` -> sampling_offsets |
 * attention_weights Linear), cross_attn.output_proj + residual + norm2 (models/transformer.py:498-506, ms_deform_attn.py:117-123,
 * :142), bbox_embed / class_embed on all decoder layers (models/lwdetr.py:149-159), enc_out_bbox_embed on the selected rows
 * (transformer.py:236-240), ref_point_head (transformer.py:28-39, :352-355).
 *   FULL stage: y = W x + b (+ `res` rows when RES) (ReLU) -> rounded -> (LayerNorm(eps) with affine -> rounded) (-> stored to `out`,
 *               row stride ldo) -> the operand of the following stages (+ `qpos` rows, rounded, when ADDQ); N = D; stage 0 may
 *               contract over k_in = 2 D input channels.
 *   SIDE stage: out[:, 0:n] = W x + b from the current operand (row stride ldo >= ceil4(n); columns up to ceil4(n) are written).
 * in (M, ld_in): k_in channels per row. wstream / vec: lwdetr_amd.kernels.pack_row_chain - 4 KB pieces (32 output channels x 64
 * k-slots in MFMA lane order), stage after stage, + 2 zero pieces; f32 vectors stage after stage: bias (D, or 32 * ceil(n / 32) for a
 * SIDE stage), then gamma (D), beta (D) when LN. */
enum { LWDETR_CHAIN_FULL = 0, LWDETR_CHAIN_SIDE = 1 };
enum { LWDETR_CHAIN_RES = 1, LWDETR_CHAIN_RELU = 2, LWDETR_CHAIN_LN = 4, LWDETR_CHAIN_STORE = 8, LWDETR_CHAIN_ADDQ = 16 };
typedef struct { int kind; int n; int flags; float eps; void* out; long ldo; } lwdetr_chain_stage;
typedef struct {
    const void* in; long ld_in; int k_in;
    const void* res; long ld_res;
    const void* qpos; long ld_q;
    const void* wstream; const float* vec;
    long M; int D; int nst;
    lwdetr_chain_stage st[6];
} lwdetr_chain_desc;
long lwdetr_row_chain_pieces(const lwdetr_chain_desc* desc);
long lwdetr_row_chain_vec_floats(const lwdetr_chain_desc* desc);
int lwdetr_row_chain(const lwdetr_chain_desc* desc, int dtype, void* hip_stream);

/* norm1 + QKV of a ViT block on its own (block 0, which has no block kernel in front of it; models/backbone/vit.py:199, :123-130):
 * q (pre-scaled by qscale), k as (B, heads, Tp, hd), v^T as (B, heads, hd, Tp) from the rows x (M, ldx). wstream / vec:
 * lwdetr_amd.kernels.pack_vit_qkv (3 C / 32 pieces of 32 features x C in natural k order, LayerNorm affine folded; f32 bias).
 * C in {192, 384}, 16-bit dtypes, M % Tp == 0, Tp % 8 == 0, hd a power of two >= 8. */
long lwdetr_vit_qkv_stream_bytes(int C);
long lwdetr_vit_qkv_vec_floats(int C);
int lwdetr_vit_qkv(const void* x, long ldx, const void* wstream, const float* vec, long M, int C, float eps, void* q_out,
                   void* k_out, void* vt_out, float qscale, int heads, int hd, int Tp, int dtype, void* hip_stream);

/* The ViT stem in one launch (round 4): x0 = patches Wpe^T + b + pos - the 16 x 16 / stride 16 patch embedding of the NCHW image plus
 * the absolute position embedding (models/backbone/vit.py:353-358, the PatchEmbed Conv2d) - stored to x (M, ldx) as the residual
 * stream, and norm1 + QKV of block 0 from the registers (vit.py:199, :123-130; outputs as lwdetr_vit_qkv). img: (B, 3, 16 Hp, 16 Wp) of
 * the model dtype; rows of x are the window-major tokens of layout (Hp, Wp, Twp) (pad rows: zero pixels), M = B * 16 * Twp; pos:
 * (16 Twp, ldpos) of the model dtype in the same token order. wstream / vec: lwdetr_amd.kernels.pack_vit_stem. Replaces the
 * PATCH16 lwdetr_gemm + lwdetr_vit_qkv pair; same rounding points (x0 rounded once from f32). C in {192, 384}, 16-bit dtypes. */
long lwdetr_vit_stem_stream_bytes(int C);
long lwdetr_vit_stem_vec_floats(int C);
int lwdetr_vit_stem(const void* img, int B, int img_h, int img_w, int Hp, int Wp, int Twp, const void* pos, long ldpos, void* x,
                    long ldx, const void* wstream, const float* vec, long M, int C, float eps, void* q_out, void* k_out,
                    void* vt_out, float qscale, int heads, int hd, int dtype, void* hip_stream);

/* ---- fused glue of the two-stage selection / decoder set-up (reference models/transformer.py:236-276, :42-68, :352-355;
 * models/lwdetr.py:150-155, :168-170). idx (B,nq) int64 = two-stage top-k; props (B,S,4) f32 anchor proposals. ---- */
int lwdetr_select_gather(const void* om, const void* enc_cls, long ldc, const float* props, const int64_t* idx,
                         void* om_sel, void* logits_out, float* props_sel, int B, int S, int d, int nq, int ncls,
                         int dtype, void* hip_stream);
/* enc_delta (B*nq,4): bbox-MLP output of the selected rows; writes enc boxes (B,nq,4), decoder reference boxes ref_out
 * (B,nq,4) f32, the (y,x,w,h) sine embedding of ref * valid_ratio[level 0] (B*nq, 2d) and the broadcast queries (B*nq, d). */
int lwdetr_decoder_inputs(const void* enc_delta, const float* props_sel, const float* refpoint, const float* valid_ratios,
                          int L, const void* query_feat, const float* dim_t, void* enc_boxes_out, float* ref_out,
                          void* sine_out, void* xdec_out, int B, int nq, int d, int dtype, void* hip_stream);
/* out[r] = (delta_xy * ref_wh + ref_xy, exp(delta_wh) * ref_wh) with ref row r % ref_rows (no sigmoid: bbox_reparam). */
int lwdetr_box_reparam(const void* delta, const float* ref, long ref_rows, void* out, long R, int dtype, void* hip_stream);
/* lwdetr_box_reparam into coord_out AND logits_out (R, ncls) contiguous = logits_pad (R rows, row stride ldc)[:, :ncls]: the
 * user-visible pred_boxes / pred_logits of all decoder layers (models/lwdetr.py:150-173) in one launch. Input row
 * layer * ref_rows + k is written to output row layer * out_layer_rows + k (out_layer_rows >= ref_rows; 0 = ref_rows, i.e.
 * contiguous): a call that owns images [b0, b0 + B) of a (layers, B_total, nq, .) tensor passes the pointers of row b0 * nq and
 * out_layer_rows = B_total * nq. */
int lwdetr_finalize_outputs(const void* delta, const float* ref, long ref_rows, void* coord_out, long R, const void* logits_pad,
                            long ldc, int ncls, void* logits_out, long out_layer_rows, int dtype, void* hip_stream);

/* ---- sorted top-k (one workgroup per image) and its two users -----------------------------------------------------
 * Order: descending value, equal values by ascending index (torch.topk leaves tie order unspecified). N < 2^20, K <= 1024.
 * lwdetr_rowmax : out[r] = max(x[r, 0:ncols]) in f32 - the class-max of the encoder logits (transformer.py:246).
 * lwdetr_topk   : x (B,N) contiguous -> idx_out (B,K) int64 (+ val_out (B,K) f32 when non-NULL); replaces
 *                 torch.topk(..., num_queries, dim=1) of the two-stage selection (transformer.py:247).
 * lwdetr_postprocess : PostProcess.forward (models/lwdetr.py:509-540): logits (B,nq,ncls) and boxes (B,nq,4 cxcywh)
 *                 contiguous in `dtype`, target_sizes (B,2) f32 rows (h,w) -> scores (B,K) f32 = sigmoid of the K largest
 *                 logits, labels (B,K) int64 = flat index % ncls, out_boxes (B,K,4) f32 xyxy in pixels. ---- */
int lwdetr_rowmax(const void* x, long ld, long rows, int ncols, float* out, int dtype, void* hip_stream);
int lwdetr_topk(const void* x, int B, int N, int K, int64_t* idx_out, float* val_out, int dtype, void* hip_stream);
int lwdetr_postprocess(const void* logits, const void* boxes, const float* target_sizes, int B, int nq, int ncls, int K,
                       float* scores, int64_t* labels, float* out_boxes, int dtype, void* hip_stream);
/* Same selection, written as the (B, K, 6) f32 record of the detection all-gather: score, label, x0, y0, x1, y1
 * (replaces the pickled per-image dicts of util/misc.py:99-139; labels < 2^24 are exact in f32). */
int lwdetr_postprocess_packed(const void* logits, const void* boxes, const float* target_sizes, int B, int nq, int ncls, int K,
                              float* packed, int dtype, void* hip_stream);

/* ---- input side: uint8 HWC -> Pillow-exact bilinear square resize -> ToTensor -> Normalize -> NCHW --
 * Replaces datasets/transforms.py:223-231 (SquareResize = PIL.Image.resize((S,S), BILINEAR)), :437-443 (Normalize) and
 * ToTensor (datasets/coco.py:127-130, deploy/benchmark.py:273-281). One descriptor per image (a DEVICE array):
 * src = uint8 RGB rows of `width` pixels, `row_stride` bytes apart; the *_off fields index `tables` (int32, device):
 * bounds = (first input index, tap count) per output column / row, coef = `ksize` 22-bit fixed-point taps per output
 * column / row (Pillow's precompute_coeffs + normalize_coeffs_8bpc, computed by the host); tmp_off = byte offset of this
 * image's height x S x 3 intermediate in `tmp`. lut = 3 x 256 floats, (v / 255 - mean[c]) / std[c]. out = (B,3,S,S). */
typedef struct {
    const uint8_t* src;
    int height, width;
    long row_stride;
    int xbounds_off, xcoef_off, xksize;
    int ybounds_off, ycoef_off, yksize;
    long tmp_off;
} lwdetr_resize_image;
int lwdetr_resize_normalize(const lwdetr_resize_image* images, int B, int max_height, const int32_t* tables, uint8_t* tmp,
                            const float* lut, void* out, int S, int dtype, void* hip_stream);

/* ---- run-time switches (tuning, A/B runs, tests). Every environment variable LWDETR_<NAME> that a launch path of this library looks at
 * (kernel choices and shapes: ATTN_LDS, ATTN_LDS_CFG, ATTN_SHORT, ATTN_WTILE, ATTN_WIN, ATTN_QT, CHAIN_SPLIT_ROWS, GEMM_BIG, GEMM_BIG_BN, GEMM_BIG_2WG,
 * CONV_PATCH, GEMM_TILE, GEMM_DMA, GEMM_KB, GEMM_NST, GEMM_PT, GEMM_PT_SKEW, MLP_SMALL_TT, FFN_SPLITS, MLP_SMALL, VB_GRID, VB_GELU16, VB_HALF, GEMM_FEW_WAVES) is read
 * ONCE per process into a table; no launch calls getenv. lwdetr_tuning_set overrides (is_set != 0) or clears (is_set == 0: back to the built-in
 * default, not to the environment) one entry by name, with or without the LWDETR_ prefix; LWDETR_ERR_BAD_ARG for an unknown name. Process-wide,
 * not synchronised with launches in flight on other threads. Results never depend on a switch beyond f32 summation order, except VB_GELU16
 * (DESIGN.md section 2). */
int lwdetr_tuning_set(const char* name, long value, int is_set);

/* ---- profiling: per-kernel HIP-event timing on the launch stream (off by default) ------------------------------ */
int lwdetr_prof_enable(int on);
int lwdetr_prof_num_kernels(void);
const char* lwdetr_prof_kernel_name(int kid);
int lwdetr_prof_collect(double* ms, double* flops, double* bytes, long long* count, int n);

#ifdef __cplusplus
}
#endif
#endif /* LWDETR_HIP_H */
