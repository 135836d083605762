"""Phase timing of lwdetr_enc_chain (tuning tool, not part of the product).

Build the instrumented library first (on the build host):   python tools/chain_timing.py --build
Run on the GPU:  LWDETR_HIP_LIB=tools/_timing/liblwdetr_hip_cht.so python tools/chain_timing.py [d k5 rows dtype]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["input loads", "first tile wait", "cv2 + LN", "value projections", "enc_output + LN", "class + max"]


def main():
    if "--build" in sys.argv:
        out = os.path.join(ROOT, "tools", "_timing")
        os.makedirs(out, exist_ok=True)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "lw-detr_amd", "csrc"), "-j8", f"OBJDIR={out}/obj_cht",
                               f"OUT={out}/liblwdetr_hip_cht.so", "TUNE=-DLWDETR_CH_TIMING=1"])
        return
    import torch
    import lwdetr_amd  # noqa: F401
    from lwdetr_amd import _native, kernels as K
    a = sys.argv[1:]
    d, k5, M = int(a[0]) if a else 256, int(a[1]) if len(a) > 1 else 640, int(a[2]) if len(a) > 2 else 51200
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[a[3] if len(a) > 3 else "fp16"]
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g)
    nl, ncls = 3, 91
    cv2 = (r(d, k5) / 25, r(d), 1 + 0.1 * r(d), 0.1 * r(d)) if k5 else None
    stream, vec = K.pack_enc_chain(d, dtype, r(d, d) / 16, r(d), 1 + 0.1 * r(d), 0.1 * r(d), r(ncls, d) / 16, r(ncls), r(nl * d, d) / 16, r(nl * d), cv2=cv2)
    dev = "cuda:0"
    x = r(M, k5 or d).to(dtype).to(dev)
    mem, om = torch.empty(M, d, dtype=dtype, device=dev), torch.empty(M, d, dtype=dtype, device=dev)
    cls, cmax = torch.empty(M, 96, dtype=dtype, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
    vals = [torch.empty(M, d, dtype=dtype, device=dev) for _ in range(nl)]
    ones = torch.ones(M, dtype=torch.uint8, device=dev)
    op = K.EncChainOp(x, k5 or d, k5, mem if k5 else None, om, cls, 96, cmax, vals, ones, ones, stream.to(dev), vec.to(dev), M=M, d=d, npix=M, S=M, lsi=0,
                      total_rows=M, ncls=ncls, eps_p=1e-6, eps_e=1e-5)
    for _ in range(3):
        op()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        e0.record(); op(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"enc chain d={d} k5={k5} rows={M} {dtype}: {ts[len(ts) // 2]:.1f} us per launch (median of 10), {(M + 127) // 128} workgroups")
    lib = _native.lib()
    if hasattr(lib, "lwdetr_debug_ch_timing"):
        buf = (C.c_ulonglong * (2 * 4 * 16))()
        assert lib.lwdetr_debug_ch_timing(buf) == 0
        for blk in range(2):
            print("workgroup", "0" if blk == 0 else "last", "(us per phase, per wave)")
            for w in range(4):
                t = [buf[(blk * 4 + w) * 16 + i] for i in range(16)]
                seg = [(t[i + 1] - t[i]) / 100.0 for i in range(6)]
                print(f"  wave {w}: " + "  ".join(f"{n} {s:.2f}" for n, s in zip(NAMES, seg)) +
                      f"  | total {(t[6] - t[0]) / 100.0:.2f}  ring waits {t[13] / 100.0:.2f} barriers {t[14] / 100.0:.2f}"
                      f"  | value stage: MFMA loops {t[11] / 100.0:.2f} epilogues {t[12] / 100.0:.2f}")


if __name__ == "__main__":
    main()
