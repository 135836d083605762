"""ctypes binding of ``liblwdetr_hip.so`` (C ABI declared in ``include/lwdetr_hip.h``).

There is deliberately NO fallback: every entry point of the product path goes through this library, and
``lib()`` raises if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C lw-detr_amd/csrc``).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LWDETR_HIP_LIB: tuning builds of the same library (tools/); the product always loads the in-tree one
LIB_PATH = os.environ.get("LWDETR_HIP_LIB") or os.path.join(_HERE, "liblwdetr_hip.so")
_lib = None

DT_F32, DT_F16, DT_BF16, DT_F64 = 0, 1, 2, 3
A_PLAIN, A_CONV3x3, A_PATCH16 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU = 0, 1, 2, 3
OUT_LINEAR, OUT_HEADS, OUT_HEADS_T, OUT_TOKMAP, OUT_DECONV2x2 = 0, 1, 2, 3, 4
ERRORS = {-1: "bad argument", -2: "unsupported configuration", -3: "kernel launch failed"}


class NativeError(RuntimeError):
    pass


class TokLayout(C.Structure):
    _fields_ = [("winmajor", C.c_int), ("Hp", C.c_int), ("Wp", C.c_int), ("Twp", C.c_int)]


class GemmSeg(C.Structure):
    _fields_ = [("out", C.c_void_p), ("out2", C.c_void_p), ("res", C.c_void_p), ("bias", C.c_void_p),
                ("gamma", C.c_void_p), ("rowmask", C.c_void_p), ("scale", C.c_float), ("act", C.c_int),
                ("mode", C.c_int), ("n_begin", C.c_int), ("n_end", C.c_int), ("ldo", C.c_long), ("ld2", C.c_long),
                ("ldres", C.c_long), ("res_mod", C.c_int), ("rowmask_after", C.c_int), ("p0", C.c_int), ("p1", C.c_int), ("p2", C.c_int),
                ("in_tok", TokLayout), ("out_tok", TokLayout), ("out_batch_stride", C.c_long),
                ("out_row_offset", C.c_long), ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p)]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("A2", C.c_void_p), ("W", C.c_void_p), ("M", C.c_int), ("N", C.c_int),
                ("K", C.c_int), ("lda", C.c_long), ("a_mode", C.c_int), ("a_tok", TokLayout),
                ("conv_cin", C.c_int), ("conv_stride", C.c_int), ("a_col0", C.c_int), ("conv_hout", C.c_int),
                ("conv_wout", C.c_int), ("img_h", C.c_int), ("img_w", C.c_int), ("nseg", C.c_int),
                ("seg", GemmSeg * 3), ("splitk_ws", C.c_void_p), ("splitk", C.c_int)]


class AttnDesc(C.Structure):
    _fields_ = [("Q", C.c_void_p), ("K", C.c_void_p), ("VT", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_long),
                ("B", C.c_int), ("heads", C.c_int), ("hd", C.c_int), ("Tp", C.c_int), ("seqs_per_img", C.c_int),
                ("seq_tok_stride", C.c_int), ("keys_per_seq", C.c_int), ("sub_stride", C.c_int),
                ("sub_len", C.c_int), ("kind", C.c_int), ("vt_slack", C.c_int)]


CHAIN_FULL, CHAIN_SIDE = 0, 1
CHAIN_RES, CHAIN_RELU, CHAIN_LN, CHAIN_STORE, CHAIN_ADDQ = 1, 2, 4, 8, 16


class ChainStage(C.Structure):
    _fields_ = [("kind", C.c_int), ("n", C.c_int), ("flags", C.c_int), ("eps", C.c_float), ("out", C.c_void_p), ("ldo", C.c_long)]


class ChainDesc(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("ld_in", C.c_long), ("k_in", C.c_int), ("res", C.c_void_p), ("ld_res", C.c_long),
                ("qpos", C.c_void_p), ("ld_q", C.c_long), ("wstream", C.c_void_p), ("vec", C.c_void_p), ("M", C.c_long),
                ("D", C.c_int), ("nst", C.c_int), ("st", ChainStage * 6)]


def is_built() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    """The loaded shared library; raises NativeError when it is missing (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not is_built():
            raise NativeError(f"lwdetr_amd: HIP extension {LIB_PATH} is missing - build it with "
                              "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                              "There is no CPU fallback for the forward path.")
        l = C.CDLL(LIB_PATH)
        vp, i, lg, f = C.c_void_p, C.c_int, C.c_long, C.c_float
        l.lwdetr_msda_forward.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, vp]
        l.lwdetr_msda_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, i, vp]
        l.lwdetr_msda_fused_forward.argtypes = [vp, vp, vp, vp, lg, i, vp, vp, vp, i, i, i, i, i, i, i, i, vp]
        l.lwdetr_gemm.argtypes = [C.POINTER(GemmDesc), i, vp]
        l.lwdetr_gemm_few.argtypes = [C.POINTER(GemmDesc), i, vp]
        l.lwdetr_attention.argtypes = [C.POINTER(AttnDesc), i, vp]
        l.lwdetr_gemm_tuning.argtypes = [i]
        l.lwdetr_gemm_tuning.restype = None
        l.lwdetr_has_experiments.argtypes = []
        l.lwdetr_gemm_pt_tuning.argtypes = [i]
        l.lwdetr_gemm_pt_tuning.restype = None
        l.lwdetr_gemm_pt_count.argtypes = []
        l.lwdetr_gemm_pt_count.restype = C.c_long
        l.lwdetr_attention_tuning.argtypes = [i]
        l.lwdetr_attention_tuning.restype = None
        l.lwdetr_attention_tuning_cfg.argtypes = [i]
        l.lwdetr_attention_tuning_cfg.restype = None
        l.lwdetr_layernorm.argtypes = [vp, lg, vp, vp, vp, lg, lg, i, f, lg, lg, lg, i, vp]
        l.lwdetr_layernorm_chain.argtypes = [vp, lg, vp, vp, f, vp, lg, vp, vp, f, vp, lg, lg, i, i, vp]
        l.lwdetr_row_stats.argtypes = [vp, lg, lg, i, f, vp, i, vp]
        l.lwdetr_ffn_splits.argtypes = [lg, i, i, i]
        l.lwdetr_ffn_partial.argtypes = [vp, lg, vp, vp, vp, vp, lg, i, i, i, vp]
        l.lwdetr_ffn_finish.argtypes = [vp, lg, vp, i, vp, vp, vp, f, vp, lg, vp, vp, f, vp, lg, lg, i, i, vp]
        l.lwdetr_mlp_fused.argtypes = [vp, lg, vp, vp, vp, vp, vp, vp, lg, vp, lg, i, f, f, vp, lg, vp, vp, vp,
                                       vp, vp, vp, vp, vp, f, i, i, i, i, vp]
        l.lwdetr_vit_block_few.argtypes = [vp, lg, vp, vp, vp, vp, vp, vp, lg, vp, lg, i, f, f, vp, lg, vp, vp, vp,
                                       vp, vp, vp, vp, vp, f, i, i, i, i, vp]
        l.lwdetr_vit_block.argtypes = [vp, lg, vp, lg, vp, vp, vp, lg, vp, lg, i, f, f, i, vp, vp, vp, f, i, i, i, i, vp]
        l.lwdetr_vit_block_stream_bytes.argtypes = [i, i]
        l.lwdetr_vit_block_stream_bytes.restype = C.c_long
        l.lwdetr_vit_block_vec_floats.argtypes = [i]
        l.lwdetr_vit_block_vec_floats.restype = C.c_long
        l.lwdetr_vit_qkv.argtypes = [vp, lg, vp, vp, lg, i, f, vp, vp, vp, f, i, i, i, i, vp]
        l.lwdetr_vit_stem.argtypes = [vp, i, i, i, i, i, i, vp, lg, vp, lg, vp, vp, lg, i, f, vp, vp, vp, f, i, i, i, vp]
        l.lwdetr_vit_stem_stream_bytes.argtypes = [i]
        l.lwdetr_vit_stem_stream_bytes.restype = C.c_long
        l.lwdetr_vit_stem_vec_floats.argtypes = [i]
        l.lwdetr_vit_stem_vec_floats.restype = C.c_long
        l.lwdetr_vit_qkv_stream_bytes.argtypes = [i]
        l.lwdetr_vit_qkv_stream_bytes.restype = C.c_long
        l.lwdetr_vit_qkv_vec_floats.argtypes = [i]
        l.lwdetr_vit_qkv_vec_floats.restype = C.c_long
        l.lwdetr_enc_chain.argtypes = [vp, lg, i, vp, vp, vp, lg, vp, vp, i, vp, vp, vp, vp, lg, i, i, i, i, lg, i, f, f, i, vp]
        l.lwdetr_enc_chain_vec_floats.argtypes = [i, i]
        l.lwdetr_enc_chain_vec_floats.restype = C.c_long
        l.lwdetr_enc_chain_pieces.argtypes = [i, i, i]
        l.lwdetr_enc_chain_pieces.restype = C.c_long
        l.lwdetr_row_chain.argtypes = [C.POINTER(ChainDesc), i, vp]
        l.lwdetr_row_chain_pieces.argtypes = [C.POINTER(ChainDesc)]
        l.lwdetr_row_chain_pieces.restype = C.c_long
        l.lwdetr_row_chain_vec_floats.argtypes = [C.POINTER(ChainDesc)]
        l.lwdetr_row_chain_vec_floats.restype = C.c_long
        l.lwdetr_select_gather.argtypes = [vp, vp, lg, vp, vp, vp, vp, vp, i, i, i, i, i, i, vp]
        l.lwdetr_decoder_inputs.argtypes = [vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, i, i, i, i, vp]
        l.lwdetr_box_reparam.argtypes = [vp, vp, lg, vp, lg, i, vp]
        l.lwdetr_rowmax.argtypes = [vp, lg, lg, i, vp, i, vp]
        l.lwdetr_topk.argtypes = [vp, i, i, i, vp, vp, i, vp]
        l.lwdetr_postprocess.argtypes = [vp, vp, vp, i, i, i, i, vp, vp, vp, i, vp]
        l.lwdetr_postprocess_packed.argtypes = [vp, vp, vp, i, i, i, i, vp, i, vp]
        l.lwdetr_finalize_outputs.argtypes = [vp, vp, lg, vp, lg, vp, lg, i, vp, lg, i, vp]
        l.lwdetr_resize_normalize.argtypes = [vp, i, i, vp, vp, vp, vp, i, i, vp]
        l.lwdetr_tuning_set.argtypes = [C.c_char_p, lg, i]
        l.lwdetr_prof_enable.argtypes = [i]
        l.lwdetr_prof_num_kernels.argtypes = []
        l.lwdetr_prof_kernel_name.argtypes = [i]
        l.lwdetr_prof_kernel_name.restype = C.c_char_p
        l.lwdetr_prof_collect.argtypes = [vp, vp, vp, vp, i]
        for fn in ("lwdetr_msda_forward", "lwdetr_msda_backward", "lwdetr_msda_fused_forward", "lwdetr_gemm", "lwdetr_attention",
                   "lwdetr_layernorm", "lwdetr_layernorm_chain", "lwdetr_row_stats", "lwdetr_enc_chain", "lwdetr_row_chain", "lwdetr_mlp_fused", "lwdetr_vit_block_few", "lwdetr_vit_block", "lwdetr_vit_qkv", "lwdetr_vit_stem", "lwdetr_ffn_splits", "lwdetr_ffn_partial", "lwdetr_ffn_finish", "lwdetr_select_gather", "lwdetr_decoder_inputs",
                   "lwdetr_box_reparam", "lwdetr_rowmax", "lwdetr_topk", "lwdetr_postprocess", "lwdetr_postprocess_packed", "lwdetr_finalize_outputs", "lwdetr_resize_normalize", "lwdetr_prof_enable", "lwdetr_prof_num_kernels", "lwdetr_prof_collect"):
            getattr(l, fn).restype = C.c_int
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"lwdetr_amd: {what} failed: {ERRORS.get(rc, rc)}")


def dtype_code(dt: torch.dtype) -> int:
    try:
        return {torch.float32: DT_F32, torch.float16: DT_F16, torch.bfloat16: DT_BF16, torch.float64: DT_F64}[dt]
    except KeyError:
        raise NativeError(f"lwdetr_amd: unsupported dtype {dt}") from None


def stream_ptr(device=None) -> int:
    """The current torch stream's hipStream_t, so kernels order with torch ops and get captured in HIP graphs."""
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeError("lwdetr_amd: tensors must live on a ROCm device (cuda:N); there is no CPU path")


# ----------------------------------------------------------------------------------------------- profiling
def tuning_set(name: str, value=None):
    """Override (value: int) or clear (None) one launch-path switch of the library (lwdetr_tuning_set; the LWDETR_* environment is read once per
    process, so tests and tools that switch inside a process go through here)."""
    check(lib().lwdetr_tuning_set(name.encode(), 0 if value is None else int(value), 0 if value is None else 1), f"tuning_set({name})")


def prof_enable(on: bool):
    check(lib().lwdetr_prof_enable(1 if on else 0), "prof_enable")


def prof_collect():
    """{kernel name: dict(ms, flops, bytes, count)} accumulated since the last call (synchronises recorded events)."""
    l = lib()
    n = l.lwdetr_prof_num_kernels()
    ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    check(l.lwdetr_prof_collect(ms, fl, by, cnt, n), "prof_collect")
    return {l.lwdetr_prof_kernel_name(k).decode(): dict(ms=ms[k], flops=fl[k], bytes=by[k], count=cnt[k])
            for k in range(n) if cnt[k]}
