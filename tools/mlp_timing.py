"""Phase timing of the fused MLP kernel inside a real forward (tuning tool, not part of the product).

Build the instrumented library first (on the build host):   python tools/mlp_timing.py --build
Run on the GPU:  LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_t1.so python tools/mlp_timing.py [--size small --batch 32]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build():
    for level in (1,):
        out = os.path.join(ROOT, "tools", "_timing")
        os.makedirs(out, exist_ok=True)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "lw-detr_amd", "csrc"), "-j8",
                               f"OBJDIR={out}/obj{level}", f"OUT={out}/liblwdetr_hip_t{level}.so",
                               f"TUNE=-DLWDETR_MLP_TIMING={level}"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--size", default="small")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dtype", default="fp16")
    a = ap.parse_args()
    if a.build:
        return build()
    import torch
    import lwdetr_amd
    from lwdetr_amd import _native, kernels as K
    from lwdetr_amd.synth import synth_images, synth_state_dict
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    dev = torch.device("cuda:0")
    model, _, _ = lwdetr_amd.build_model(lwdetr_amd.get_args(a.size))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).to(dt).eval()
    x = synth_images(a.batch, 640, 640, seed=1).to(dev).to(dt)
    for _ in range(3):
        model(x)
    plan = model._plan(a.batch, 640, 640)
    stream = _native.stream_ptr(dev)
    buf = (C.c_ulonglong * (8 * 16))()
    names = ["prologue(proj)", "layernorm", "hidden loop", "epilogue", "qkv exchange", "qkv"]
    n = 0
    for op in plan.ops_backbone:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        op(stream)
        e1.record()
        torch.cuda.synchronize()
        if isinstance(op, K.MlpFusedOp):
            assert _native.lib().lwdetr_debug_mlp_timing(buf) == 0
            t = [[buf[w * 16 + i] for i in range(16)] for w in range(8)]
            print(f"mlp launch {n}: {e0.elapsed_time(e1) * 1e3:.1f} us (workgroup 0, cycles per wave)")
            for w in range(8):
                seg = [t[w][i + 1] - t[w][i] for i in range(6)]
                line = "  wave %d: " % w + "  ".join(f"{nm} {s}" for nm, s in zip(names, seg)) + f"  total {t[w][6] - t[w][0]}"
                print(line)
            n += 1


if __name__ == "__main__":
    main()
