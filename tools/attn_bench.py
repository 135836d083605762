"""Times the global-attention launch shapes of the BASELINE configs with the attn_kernel / LDS-ring variants
(LWDETR_ATTN_LDS, LWDETR_ATTN_LDS_CFG are read once per process, so every variant runs in a child process).

    python tools/attn_bench.py            # all shapes x all variants
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = {  # name: (B, heads, hd, twp, tw, dtype)
    "small_b32_f16": (32, 12, 16, 100, 100, "float16"),
    "medium_b64_bf16": (64, 12, 32, 100, 100, "bfloat16"),
    "large_b32_f16": (32, 12, 32, 100, 100, "float16"),
    "xlarge960_b16_f16": (16, 12, 64, 228, 225, "float16"),
    "small_b1_f16": (1, 12, 16, 100, 100, "float16"),
    # window attention (16 sequences per image): name suffix _win
    "small_b32_f16_win": (32, 12, 16, 100, 100, "float16"),
    "medium_b64_bf16_win": (64, 12, 32, 100, 100, "bfloat16"),
    "xlarge960_b16_f16_win": (16, 12, 64, 228, 225, "float16"),
    "small_b16_f16_win": (16, 12, 16, 100, 100, "float16"),
    "large_b32_f16_win": (32, 12, 32, 100, 100, "float16"),
}
VARIANTS = [("attn_kernel", {"LWDETR_ATTN_LDS": "0", "LWDETR_ATTN_WIN": "0", "LWDETR_ATTN_SHORT": "0", "LWDETR_ATTN_WTILE": "0"}),
            ("attn_kernel, loads up front short", {"LWDETR_ATTN_LDS": "0", "LWDETR_ATTN_WIN": "0", "LWDETR_ATTN_WTILE": "0"}),
            ("one wave / window win", {"LWDETR_ATTN_WIN": "1", "LWDETR_ATTN_WTILE": "0"}),
            ("window tile through LDS wtile", {"LWDETR_ATTN_WTILE": "1"}),
            ("lds 1x8", {"LWDETR_ATTN_LDS_CFG": "108", "LWDETR_ATTN_WIN": "0"}),
            ("lds 1x10", {"LWDETR_ATTN_LDS_CFG": "110"}), ("lds 2x4", {"LWDETR_ATTN_LDS_CFG": "204"}),
            ("lds 2x5", {"LWDETR_ATTN_LDS_CFG": "205"}), ("lds 2x8", {"LWDETR_ATTN_LDS_CFG": "208"}),
            # 128 / 256 keys per ring step (round 4): 1000 (U - 1) + 100 QT + NW
            ("lds 1x8u2", {"LWDETR_ATTN_LDS_CFG": "1108"}), ("lds 1x10u2", {"LWDETR_ATTN_LDS_CFG": "1110"}),
            ("lds 1x4", {"LWDETR_ATTN_LDS_CFG": "104", "LWDETR_ATTN_WIN": "0"}), ("lds 1x5", {"LWDETR_ATTN_LDS_CFG": "105"}),
            ("lds 1x2", {"LWDETR_ATTN_LDS_CFG": "102"}), ("lds 1x3", {"LWDETR_ATTN_LDS_CFG": "103"}), ("lds 1x6", {"LWDETR_ATTN_LDS_CFG": "106"}),
            ("lds 1x4u2", {"LWDETR_ATTN_LDS_CFG": "1104"}), ("lds 1x5u2", {"LWDETR_ATTN_LDS_CFG": "1105"}),
            ("lds 1x8u4", {"LWDETR_ATTN_LDS_CFG": "3108"}), ("lds 2x8u2", {"LWDETR_ATTN_LDS_CFG": "1208"})]


def child(shape):
    import torch
    from lwdetr_amd import kernels as K
    B, heads, hd, twp, tw, dt = SHAPES[shape]
    T = getattr(torch, dt)
    dev, Tp = "cuda:0", 16 * twp
    g = torch.Generator(device="cpu").manual_seed(0)
    q = (torch.randn(B, heads, Tp, hd, generator=g) * 0.5).to(dev).to(T)
    k = torch.randn(B, heads, Tp, hd, generator=g).to(dev).to(T)
    store = torch.zeros(B * heads * hd * Tp + 8, device=dev, dtype=T)
    vt = store[:B * heads * hd * Tp].view(B, heads, hd, Tp)
    vt.copy_(torch.randn(B, heads, hd, Tp, generator=g).to(T))
    out = torch.zeros(B * Tp, heads * hd, device=dev, dtype=T)
    win = shape.endswith("_win")
    op = K.AttnOp(q, k, vt, out, B=B, heads=heads, hd=hd, Tp=Tp, ldo=heads * hd, seqs_per_img=16 if win else 1,
                  seq_tok_stride=twp if win else Tp, keys_per_seq=twp if win else Tp, sub_stride=twp, sub_len=tw,
                  kind=0 if win else 1, vt_slack=True)
    for _ in range(3):
        op()
    err = ""
    if os.environ.get("ATTN_BENCH_CHECK") == "1" and not win and tw == twp:       # against torch in f32 (global attention, no pad rows)
        ref = torch.softmax(q.float() @ k.float().transpose(-1, -2) * 0.6931471805599453, -1) @ vt.float().transpose(-1, -2)   # q carries hd^-0.5 log2(e): the kernel works in base 2
        got = out.view(B, Tp, heads, hd).permute(0, 2, 1, 3).float()
        err = f"  max|d| vs torch f32 {float((got - ref).abs().max()):.2e}"
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); op(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    flops = 4.0 * B * heads * Tp * (twp if win else Tp) * hd
    med = ts[len(ts) // 2]
    print(f"{med:9.1f} us (min {ts[0]:.1f})  {flops / med / 1e6:7.1f} TFLOP/s{err}", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    only = [a for a in sys.argv[1:] if not a.startswith("--v=")] or list(SHAPES)
    vsel = [a[4:].split(",") for a in sys.argv[1:] if a.startswith("--v=")]
    variants = [v for v in VARIANTS if not vsel or v[0].split()[-1] in vsel[0]]
    for shape in only:
        for name, env in variants:
            e = dict(os.environ, **env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", shape], env=e, capture_output=True,
                               text=True, timeout=300)
            res = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else f"FAILED rc={r.returncode} {r.stderr[-300:]}"
            print(f"{shape:20s} {name:12s} {res}", flush=True)


if __name__ == "__main__":
    main()
