"""Old (lwdetr_mlp_fused) vs new (lwdetr_vit_block) fused ViT block kernel on one shape (tuning tool, not part of the product).

python tools/vitblock_bench.py [C] [batch] [dtype] [iters]     default: 192 32 fp16 20  (BASELINE config 2: M = 51200)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lwdetr_amd import kernels as K  # noqa: E402


def main():
    c = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[3] if len(sys.argv) > 3 else "fp16"]
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    tp, heads = 1600, c // 32
    hd, m = c // heads, batch * 1600
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    x, att = (r(m, c) * 2).to(dtype).to(dev), r(m, c).to(dtype).to(dev)
    w1, b1, w2, b2 = r(4 * c, c, sc=c ** -0.5), r(4 * c, sc=0.1), r(c, 4 * c, sc=(4 * c) ** -0.5), r(c, sc=0.1)
    lw, lb, g1, g2 = r(c, sc=0.2) + 1, r(c, sc=0.1), r(c, sc=0.05) + 0.3, r(c, sc=0.05) + 0.3
    wp, bp = r(c, c, sc=c ** -0.5), r(c, sc=0.1)
    wqkv, qb, vb, lw1, lb1 = r(3 * c, c, sc=c ** -0.5), r(c, sc=0.1), r(c, sc=0.1), r(c, sc=0.2) + 1, r(c, sc=0.1)
    q = torch.zeros(batch, heads, tp, hd, dtype=dtype, device=dev)
    k, vt = torch.zeros_like(q), torch.zeros(batch, heads, hd, tp, dtype=dtype, device=dev)
    taps = torch.zeros(m, c, dtype=dtype, device=dev)
    todev = lambda *ts: [t.to(dev) for t in ts]
    stream, vec = todev(*K.pack_vit_block(wp, bp, g1, w1, b1, w2, b2, g2, lw, lb, dtype, qkv=(wqkv, qb, vb, lw1, lb1)))
    w1p, b1p, w2p = todev(*K.pack_mlp_weights(w1.to(dev), b1.to(dev), w2.to(dev), lw.to(dev), lb.to(dev), dtype, proj=True))
    wq, bq = todev(*K.pack_qkv_weights(wqkv, qb, vb, lw1, lb1, dtype))
    kw = dict(q=q, k=k, vt=vt, qscale=0.25, heads=heads, hd=hd, Tp=tp)
    xs = [x.clone() for _ in range(2)]
    new = K.VitBlockOp(xs[0], att, stream, vec, m, c, 1e-6, out2=taps, ld2=c, **kw)
    old = K.MlpFusedOp(xs[1], w1p, b1p, w2p, b2.to(dev), g2.to(dev), m, c, 1e-6, out2=taps, ld2=c, att=att, wp=wp.to(dtype).to(dev).contiguous(),
                       bp=bp.to(dev), gamma1=g1.to(dev), wqkv=wq, bqkv=bq, **kw)
    flops = 24.0 * m * c * c
    for name, op, buf in (("vit_block", new, xs[0]), ("mlp_fused", old, xs[1])):
        if os.environ.get("ONLY") and os.environ["ONLY"] != name:
            continue
        ts = []
        for it in range(iters + 3):
            buf.copy_(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); op(); e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{name}: C={c} M={m} {dtype}: median {med:.1f} us  min {ts[0]:.1f}  max {ts[-1]:.1f}  -> {flops / med / 1e6:.0f} TFLOP/s ({flops / med / 1e6 / 2500:.3f} of MFMA peak)")


if __name__ == "__main__":
    main()
