"""Ablation timing of the global-attention kernel (tuning tool): builds private copies of the library with parts of the
inner loop removed (-DLWDETR_ATTN_ABL=n, wrong results) and times one launch shape. --build on the build host first."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = [0, 1, 2, 3, 4, 8, 12, 15, 31]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--variant", type=int, default=None)
    a = ap.parse_args()
    out = os.path.join(ROOT, "tools", "_timing")
    if a.build:
        for v in VARIANTS:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "lw-detr_amd", "csrc"), "-j8", f"OBJDIR={out}/abl{v}",
                                   f"OUT={out}/liblwdetr_hip_abl{v}.so", f"TUNE=-DLWDETR_ATTN_ABL={v}"])
        return
    if a.variant is None:
        for v in VARIANTS:
            env = dict(os.environ, LWDETR_HIP_LIB=f"{out}/liblwdetr_hip_abl{v}.so")
            subprocess.call([sys.executable, __file__, "--variant", str(v)], env=env)
        return
    import torch
    from lwdetr_amd import kernels as K
    dev = "cuda:0"
    B, heads, hd, Tp = 32, 12, 16, 1600
    q = torch.randn(B, heads, Tp, hd, device=dev).half() * 0.3
    k = torch.randn(B, heads, Tp, hd, device=dev).half()
    vt = torch.randn(B, heads, hd, Tp, device=dev).half()
    out_t = torch.zeros(B * Tp, heads * hd, device=dev).half()
    op = K.AttnOp(q, k, vt, out_t, B=B, heads=heads, hd=hd, Tp=Tp, ldo=heads * hd, seqs_per_img=1, seq_tok_stride=Tp,
                  keys_per_seq=Tp, sub_stride=100, sub_len=100, kind=1)
    for _ in range(3):
        op()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); op(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    names = {1: "no exp2", 2: "no max", 4: "no QK mfma", 8: "no PV/sum mfma", 16: "no cvt"}
    what = " + ".join(n for b, n in names.items() if a.variant & b) or "full kernel"
    print(f"variant {a.variant:2d} ({what}): {ts[len(ts) // 2]:.1f} us")


if __name__ == "__main__":
    main()
