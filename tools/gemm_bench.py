"""Micro-benchmark of the fused GEMM / attention / LayerNorm kernels on the shapes of one LW-DETR config (GPU only)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd import kernels as K  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--c", type=int, default=192)
    ap.add_argument("--dtype", default="fp16")
    a = ap.parse_args()
    T = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a.dtype]
    dev = "cuda:0"
    M, C = a.batch * 1600, a.c
    es = 2 if T != torch.float32 else 4
    print(f"M={M} C={C} dtype={a.dtype}")
    for name, n, k, act, res in [("qkv", 3 * C, C, K.ACT_NONE, False), ("proj", C, C, K.ACT_NONE, True),
                                 ("fc1", 4 * C, C, K.ACT_GELU, False), ("fc1-noact", 4 * C, C, K.ACT_NONE, False),
                                 ("fc2", C, 4 * C, K.ACT_NONE, True), ("cv1", 256, 4 * C, K.ACT_SILU, False),
                                 ("cls", 91, 256, K.ACT_NONE, False)]:
        x = torch.randn(M, k, device=dev).to(T)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(T)
        b = torch.randn(n, device=dev)
        ld = (n + 3) // 4 * 4
        out = torch.empty(M, ld, device=dev, dtype=T)
        r = torch.randn(M, n, device=dev).to(T) if res else None
        g = torch.rand(n, device=dev) if res else None
        op = K.GemmOp(x, w, M, n, k, [K.seg(out, 0, n, ldo=ld, bias=b, act=act, res=r, ldres=n, gamma=g)])
        us = timeit(op)
        fl = 2.0 * M * n * k
        by = (M * k + n * k + M * n * (2 if res else 1)) * es
        print(f"{name:10s} N={n:4d} K={k:4d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:7.1f} GB/s (alg)")
    # attention
    heads, hd = 12, C // 12
    tp = 1600
    q = torch.randn(a.batch, heads, tp, hd, device=dev).to(T)
    kk = torch.randn(a.batch, heads, tp, hd, device=dev).to(T)
    vt = torch.randn(a.batch, heads, hd, tp, device=dev).to(T)
    o = torch.empty(a.batch * tp, C, device=dev, dtype=T)
    for name, spi, keys, kind in [("attn_window", 16, 100, 0), ("attn_global", 1, 1600, 1)]:
        op = K.AttnOp(q, kk, vt, o, B=a.batch, heads=heads, hd=hd, Tp=tp, ldo=C, seqs_per_img=spi,
                      seq_tok_stride=100 if spi == 16 else tp, keys_per_seq=keys, sub_stride=100, sub_len=100, kind=kind)
        us = timeit(op)
        fl = 4.0 * a.batch * spi * heads * keys * keys * hd
        print(f"{name:12s}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {4 * M * C * es / us / 1e3:7.1f} GB/s (alg)")
    x = torch.randn(M, C, device=dev).to(T)
    gm, bt = torch.rand(C, device=dev), torch.rand(C, device=dev)
    op = K.LayerNormOp(x, gm, bt, torch.empty_like(x), M, C, 1e-6)
    us = timeit(op)
    print(f"layernorm C={C}: {us:8.1f} us  {2 * M * C * es / us / 1e3:7.1f} GB/s")
    # reference points: hipBLASLt through torch, and a device copy
    x = torch.randn(M, C, device=dev).to(T)
    w = torch.randn(4 * C, C, device=dev).to(T)
    us = timeit(lambda: torch.nn.functional.linear(x, w))
    print(f"torch F.linear fc1 (hipBLASLt, no epilogue): {us:8.1f} us")
    big = torch.empty(256 << 20, device=dev, dtype=torch.uint8)
    dst = torch.empty_like(big)
    us = timeit(lambda: dst.copy_(big))
    print(f"device copy 256 MiB: {us:8.1f} us = {2 * big.numel() / us / 1e3:7.1f} GB/s")


def bench_mlp(batch=32, C=192, dtype=torch.float16):
    dev = "cuda:0"
    M = batch * 1600
    x = torch.randn(M, C, device=dev).to(dtype)
    w1, b1 = torch.randn(4 * C, C) * C ** -0.5, torch.randn(4 * C) * 0.1
    w2, b2 = torch.randn(C, 4 * C) * (4 * C) ** -0.5, torch.randn(C, device=dev) * 0.1
    lw, lb, g2 = torch.rand(C) + 0.5, torch.randn(C) * 0.1, torch.rand(C, device=dev) * 0.3
    w1f, b1f, w2c = (t.to(dev) for t in K.pack_mlp_weights(w1, b1, w2, lw, lb, dtype))
    op = K.MlpFusedOp(x, w1f, b1f, w2c, b2, g2, M, C, 1e-6)
    us0 = timeit(op)
    att = torch.randn(M, C, device=dev).to(dtype); wp = (torch.randn(C, C, device=dev) * C ** -0.5).to(dtype)
    w1p, b1p, w2p = (t.to(dev) for t in K.pack_mlp_weights(w1, b1, w2, lw, lb, dtype, proj=True))
    op = K.MlpFusedOp(x, w1p, b1p, w2p, b2, g2, M, C, 1e-6, att=att, wp=wp, bp=b2, gamma1=g2)
    print(f"mlp_fused (no proj) C={C}: {us0:8.1f} us")
    us = timeit(op)
    print(f"mlp_fused C={C}: {us:8.1f} us  {16.0 * M * C * C / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
    bench_mlp()


