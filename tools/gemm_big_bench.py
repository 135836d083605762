"""Large-tile GEMM kernel vs the 64 x 64 / 128 x 128 ring kernels on the ViT GEMM shapes of the C = 384 / 768 models.

    python tools/gemm_big_bench.py [xlarge|medium|large ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd import _native, kernels as K  # noqa: E402

SETS = {
    "xlarge": (16 * 3648, 768, torch.float16),
    "medium": (64 * 1600, 384, torch.bfloat16),
    "large": (32 * 1600, 384, torch.float16),
    "small": (32 * 1600, 192, torch.float16),
}


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = "cuda:0"
    lib = _native.lib()
    for name in (sys.argv[1:] or ["xlarge", "medium", "large"]):
        M, C, T = SETS[name]
        print(f"== {name}: M={M} C={C} {T}")
        for what, n, k, act, res in [("qkv", 3 * C, C, K.ACT_NONE, False), ("proj", C, C, K.ACT_NONE, True),
                                     ("fc1", 4 * C, C, K.ACT_GELU, False), ("fc2", C, 4 * C, K.ACT_NONE, True)]:
            x = torch.randn(M, k, device=dev).to(T)
            w = (torch.randn(n, k, device=dev) * k ** -0.5).to(T)
            b = torch.randn(n, device=dev)
            out = torch.empty(M, n, device=dev, dtype=T)
            r = torch.randn(M, n, device=dev).to(T) if res else None
            g = torch.rand(n, device=dev) if res else None
            op = K.GemmOp(x, w, M, n, k, [K.seg(out, 0, n, ldo=n, bias=b, act=act, res=r, ldres=n, gamma=g)])
            fl = 2.0 * M * n * k
            row = f"{what:5s} N={n:5d} K={k:5d}:"
            ref = None
            for mode, label in [(m_, l_) for m_, l_ in [(64, "big kb64"), (32, "big kb32"), (128, "4w128 2wg")]
                                if str(m_) in os.environ.get("GEMM_BENCH_MODES", "64,32,128").split(",")]:
                lib.lwdetr_gemm_tuning(mode)
                us = timeit(op)
                if ref is None:
                    ref = out.float().clone()
                    err = 0.0
                else:
                    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
                row += f"  {label} {us:8.1f} us {fl / us / 1e6:7.1f} TF/s (rel diff {err:.1e})"
            lib.lwdetr_gemm_tuning(-1)
            print(row, flush=True)


if __name__ == "__main__":
    main()
