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
        for what, n, k, act, res in [("qkv", 3 * C, C, K.ACT_NONE, False), ("qkvh", 3 * C, C, K.ACT_NONE, False), ("proj", C, C, K.ACT_NONE, True),
                                     ("fc1", 4 * C, C, K.ACT_GELU, False), ("fc2", C, 4 * C, K.ACT_NONE, True)]:
            x = torch.randn(M, k, device=dev).to(T)
            w = (torch.randn(n, k, device=dev) * k ** -0.5).to(T)
            b = torch.randn(n, device=dev)
            out = torch.empty(M, n, device=dev, dtype=T)
            r = torch.randn(M, n, device=dev).to(T) if res else None
            g = torch.rand(n, device=dev) if res else None
            if what == "qkvh":          # the model's QKV launch: q / k as (B, heads, Tp, hd), v transposed
                heads, hd = 12, C // 12
                tp = 3648 if name == "xlarge" else 1600
                q_, k_, vt_ = out[:, :C], out[:, C:2 * C], out[:, 2 * C:]
                q_ = out.view(-1)[:M * C].view(M // tp, heads, tp, hd)
                k_ = out.view(-1)[M * C:2 * M * C].view(M // tp, heads, tp, hd)
                vt_ = out.view(-1)[2 * M * C:].view(M // tp, heads, hd, tp)
                op = K.GemmOp(x, w, M, n, k, [K.seg(q_, 0, C, mode=K.OUT_HEADS, bias=b[:C].contiguous(), scale=0.18, p0=tp, p1=hd, p2=heads),
                                              K.seg(k_, C, 2 * C, mode=K.OUT_HEADS, p0=tp, p1=hd, p2=heads),
                                              K.seg(vt_, 2 * C, 3 * C, mode=K.OUT_HEADS_T, bias=b[2 * C:].contiguous(), p0=tp, p1=hd, p2=heads)])
            else:
                op = K.GemmOp(x, w, M, n, k, [K.seg(out, 0, n, ldo=n, bias=b, act=act, res=r, ldres=n, gamma=g)])
            fl = 2.0 * M * n * k
            row = f"{what:5s} N={n:5d} K={k:5d}:"
            # "pt" = the persistent large-tile kernel of round 6 (gemm_pt.hip); the numeric modes are gemm_big_kernel's stage depths / forms
            modes = [(m_, l_) for m_, l_ in [(64, "big kb64"), ("pt", "persistent"), (32, "big kb32"), (128, "4w128 2wg")]
                     if str(m_) in os.environ.get("GEMM_BENCH_MODES", "64,pt").split(",")]
            modes += [(f"pt{w_}", f"pt skew {w_} us") for w_ in os.environ.get("PT_SKEWS", "").split(",") if w_]      # start-skew windows in us (tuning)

            def select(mode):
                is_pt = str(mode).startswith("pt")
                arg = str(mode)[2:]
                lib.lwdetr_gemm_pt_tuning((2 + 256 * int(float(arg) * 100) if arg else 2) if is_pt else 0)
                lib.lwdetr_gemm_tuning(64 if is_pt else mode)

            # interleaved rounds (the modes take turns; a single pass per mode drifts by +-5 % with the clock): median per mode
            times = {m_: [] for m_, _ in modes}
            outs = {}
            for rnd in range(int(os.environ.get("GEMM_BENCH_ROUNDS", "5"))):
                for mode, _ in modes:
                    select(mode)
                    times[mode].append(timeit(op, iters=4))
                    if rnd == 0:
                        outs[mode] = out.float().clone()
            ref = outs[modes[0][0]]
            for mode, label in modes:
                ts = sorted(times[mode])
                us = ts[len(ts) // 2]
                err = ((outs[mode] - ref).abs().max() / ref.abs().max()).item()
                row += f"  {label} {us:7.1f} us [{ts[0]:.0f}-{ts[-1]:.0f}] {fl / us / 1e6:6.1f} TF/s (diff {err:.0e})"
            lib.lwdetr_gemm_tuning(-1)
            lib.lwdetr_gemm_pt_tuning(-1)
            print(row, flush=True)


if __name__ == "__main__":
    main()
