"""Time the two launches of the decoder FFN (lwdetr_ffn_partial, lwdetr_ffn_finish) and the unfused three-launch plan.
LWDETR_FFN_SPLITS caps the hidden split (read once per process): run once per value."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lwdetr_amd  # noqa: E402
from lwdetr_amd import _native as N, kernels as K  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c", type=int, default=256)
    ap.add_argument("--hid", type=int, default=2048)
    ap.add_argument("--rows", type=int, nargs="+", default=[300, 1200, 9600, 19200])
    a = ap.parse_args()
    dev, dt = "cuda", torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    c, hid = a.c, a.hid
    w1, b1, w2, b2 = r(hid, c) * c ** -0.5, r(hid) * 0.1, r(c, hid) * hid ** -0.5, r(c) * 0.1
    g1, be1, g2, be2 = r(c) * 0.1 + 1, r(c) * 0.1, r(c) * 0.1 + 1, r(c) * 0.1
    w1p, b1p, w2c = K.pack_mlp_weights(w1, b1, w2, None, None, dt)
    for m in a.rows:
        x = r(m, c).to(dt)
        o1, o2 = torch.empty_like(x), torch.empty_like(x)
        op = K.FfnOp(x, w1p, b1p, w2c, b2, g1, be1, 1e-5, o1, g2, be2, 1e-5, o2, m, c)
        st = N.stream_ptr()
        t_part = timeit(lambda: op._f_part(*op.a_part, st))
        t_fin = timeit(lambda: op._f_fin(*op.a_fin, st))
        t_both = timeit(op)
        h, y = torch.empty(m, hid, dtype=dt, device=dev), torch.empty_like(x)
        g_1 = K.GemmOp(x, w1.to(dt), m, hid, c, [K.seg(h, 0, hid, ldo=hid, bias=b1, act=K.ACT_RELU)])
        g_2 = K.GemmOp(h, w2.to(dt), m, c, hid, [K.seg(y, 0, c, ldo=c, bias=b2, res=x, ldres=c)])
        ln = K.LayerNormChainOp(y, g1, be1, 1e-5, o1, g2, be2, 1e-5, o2, m, c)
        t_un = timeit(lambda: (g_1(), g_2(), ln()))
        print(f"M={m:6d} C={c} hid={hid} splits={op.splits:2d}  partial {t_part:6.1f} us  finish {t_fin:6.1f} us  both {t_both:6.1f} us"
              f"  | unfused 3 launches {t_un:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
