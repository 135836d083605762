"""Time the 3x3 convolution launch of the C2f bottlenecks (tuning tool): python tools/conv_time.py [batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lwdetr_amd import kernels as K


def problem(b, hp, wp, c, seed, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(b * hp * wp, 5 * c, generator=g) * 0.7).to(dtype).cuda()
    w = (torch.randn(c, 9 * c, generator=g) * (9 * c) ** -0.5).to(dtype).cuda()
    bias = torch.randn(c, generator=g).cuda()
    out = torch.zeros(b * hp * wp, c, dtype=dtype, device="cuda")
    return K.GemmOp(x, w, b * hp * wp, c, 9 * c, [K.seg(out, 0, c, ldo=c, bias=bias, act=K.ACT_SILU)], lda=5 * c, a_mode=K.A_CONV3x3,
                    a_tok=K.tok_layout(False, hp, wp, 0), conv_cin=c, conv_stride=1, a_col0=2 * c, conv_hout=hp, conv_wout=wp, keep=(out, bias))


def main():
    c = int(os.environ.get("CONV_C", "128")); hw = int(os.environ.get("CONV_HW", "40"))
    for b in [int(v) for v in sys.argv[1:]] or [16, 32]:
        op = problem(b, hw, hw, c, 1)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            op(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(5):
            e0.record()
            for _ in range(20):
                op(st)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        if os.environ.get("CONV_TIMING"):
            import ctypes
            from lwdetr_amd import _native
            buf = (ctypes.c_ulonglong * 48)()
            op(st); torch.cuda.synchronize()
            f = _native.lib().lwdetr_debug_conv_timing; f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int
            assert f(ctypes.cast(buf, ctypes.c_void_p)) == 0
            names = ["issue DMA", "masks", "wait all + barrier", "k-loop", "epilogue", "  loop: lgkmcnt wait", "  loop: vmcnt wait", "  loop: barrier", "kernel"]
            for w in range(4):
                print(f"   wave {w}: " + ", ".join(f"{n} {buf[12 * w + i] / 100:.2f} us" for i, n in enumerate(names)), flush=True)
        m = b * hw * hw
        print(f"B {b:3d} ({m} rows, {hw}x{hw}, C {c}): {best:7.1f} us  {2 * m * c * 9 * c / best / 1e6:6.0f} TFLOP/s", flush=True)


main()
