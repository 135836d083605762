"""Phase timing of the large-tile GEMM kernel (tuning tool, not part of the product).

Build the instrumented libraries first (on the build host):   python tools/big_timing.py --build
Run on the GPU:  python tools/big_timing.py
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WG = 100


def lib_path(sched):
    return os.path.join(ROOT, "tools", "_timing", f"libbig_t{sched}.so")


def build():
    out = os.path.join(ROOT, "tools", "_timing")
    os.makedirs(out, exist_ok=True)
    for sched in (1,):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "lw-detr_amd", "csrc"), "-j8", f"OBJDIR={out}/objbig_t{sched}",
                               f"OUT={lib_path(sched)}", f"TUNE=-DLWDETR_BIG_TIMING={WG}"])


def main():
    if "--build" in sys.argv:
        return build()
    if "LWDETR_HIP_LIB" not in os.environ:       # one process per instrumented library (the library is loaded once)
        for sched in (1,):
            subprocess.check_call([sys.executable, __file__], env=dict(os.environ, LWDETR_HIP_LIB=lib_path(sched)))
        return
    import torch
    from lwdetr_amd import _native, kernels as K
    lib = _native.lib()
    lib.lwdetr_debug_big_timing.argtypes = [C.c_void_p]
    print("==", os.path.basename(os.environ["LWDETR_HIP_LIB"]))
    dev, T = "cuda:0", torch.float16
    for M, n, k in [(58368, 768, 3072), (58368, 2304, 768), (58368, 3072, 768), (51200, 384, 1536), (51200, 1536, 384)]:
        x = torch.randn(M, k, device=dev).to(T)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(T)
        out = torch.empty(M, n, device=dev, dtype=T)
        for mode in (64, 32):
            lib.lwdetr_gemm_tuning(mode)
            op = K.GemmOp(x, w, M, n, k, [K.seg(out, 0, n, ldo=n)])
            for _ in range(3):
                op()
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * 64)()
            assert lib.lwdetr_debug_big_timing(buf) == 0
            t = [list(buf[8 * wv:8 * wv + 8]) for wv in range(8)]
            nk = t[0][5]
            row = f"M={M} N={n} K={k} kb{mode}: steps {nk}"
            # 10 ns ticks -> ns per step
            f = lambda v: 10.0 * v / max(nk, 1)
            row += "  per step [ns] wait " + "/".join(f"{f(t[wv][0]):.0f}" for wv in range(8))
            row += "  barrier " + "/".join(f"{f(t[wv][1]):.0f}" for wv in range(8))
            row += "  mma " + "/".join(f"{f(t[wv][2]):.0f}" for wv in range(8))
            row += f"  | epilogue {10 * t[0][3]} ns, tile total {10 * t[0][4]} ns"
            for wv in (0, 7):
                v = t[wv][6]
                row += f"  [wave {wv}: drain {10 * (v & 0xffff)} stage-sync {10 * ((v >> 16) & 0xffff)} finish {10 * ((v >> 32) & 0xffff)} end-sync {10 * ((v >> 48) & 0xffff)} ns]"
            print(row, flush=True)


if __name__ == "__main__":
    main()
