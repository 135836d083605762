"""Phase timing of lwdetr_vit_block (tuning tool, not part of the product).

NOTE (end of round 6): since vitblock.o is built without SLP vectorisation the INSTRUMENTED kernel spills far more than the product kernel (192 against 24
bytes of scratch per lane; its LayerNorm phase reads 8.6 us instead of ~3.5) - the stamps no longer describe the product kernel's phases one to one
(profiles/r6g_*). The ablation builds compare like with like and remain usable.

Build the instrumented library first (on the build host):   python tools/vitblock_timing.py --build
Run on the GPU:  LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_vbt.so python tools/vitblock_timing.py [C batch dtype]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["first loads + vec", "acc init", "projection", "x1 + LayerNorm", "pre-step fc1(0)", "iteration 0", "hidden loop", "last iteration",
         "post-step fc2", "epilogue", "QKV"]


def main():
    if "--build" in sys.argv:
        out = os.path.join(ROOT, "tools", "_timing")
        os.makedirs(out, exist_ok=True)
        for abl in [0] + [int(a) for a in sys.argv[2:] if a.isdigit()]:       # --build [ablation bits ...]
            tag = "" if abl == 0 else f"_a{abl}"
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "lw-detr_amd", "csrc"), "-j8", f"OBJDIR={out}/obj_vbt{tag}",
                                   f"OUT={out}/liblwdetr_hip_vbt{tag}.so", f"TUNE=-DLWDETR_VB_TIMING=1 -DLWDETR_VB_ABLATE={abl}"])
        return
    import torch
    sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:] if not a.startswith("--")]
    os.environ["ONLY"] = "vit_block"
    import tools.vitblock_bench as B
    from lwdetr_amd import _native
    B.main()
    buf = (C.c_ulonglong * (2 * 4 * 16))()
    assert _native.lib().lwdetr_debug_vb_timing(buf) == 0
    for blk in range(2):
        print("workgroup", "0" if blk == 0 else "last", "(us per phase, per wave)")
        for w in range(4):
            t = [buf[(blk * 4 + w) * 16 + i] for i in range(16)]
            seg = [(t[i + 1] - t[i]) / 100.0 for i in range(11)]
            print(f"  wave {w}: " + "  ".join(f"{n} {s:.2f}" for n, s in zip(NAMES, seg)) +
                  f"  | total {(t[11] - t[0]) / 100.0:.2f}  loop waits {t[13] / 100.0:.2f} barriers {t[14] / 100.0:.2f}"
                  f"  shader clock {(t[15] - t[12]) / max(t[11] - t[0], 1) * 100.0:.0f} MHz")


if __name__ == "__main__":
    main()
