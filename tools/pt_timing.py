"""Phase timing of the persistent large-tile GEMM kernel, gemm_pt.hip (tuning tool, not part of the product).

Build the instrumented library first (on the build host):   python tools/pt_timing.py --build
Run on the GPU:  python tools/pt_timing.py
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WG = 100
LIB = os.path.join(ROOT, "tools", "_timing", "libpt_t.so")


def build():
    out = os.path.join(ROOT, "tools", "_timing")
    os.makedirs(out, exist_ok=True)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "lw-detr_amd", "csrc"), "-j8", f"OBJDIR={out}/objpt_t",
                           f"OUT={LIB}", f"TUNE=-DLWDETR_PT_TIMING={WG}"])


def main():
    if "--build" in sys.argv:
        return build()
    if "LWDETR_HIP_LIB" not in os.environ:
        subprocess.check_call([sys.executable, __file__], env=dict(os.environ, LWDETR_HIP_LIB=LIB))
        return
    import torch
    from lwdetr_amd import _native, kernels as K
    lib = _native.lib()
    lib.lwdetr_debug_pt_timing.argtypes = [C.c_void_p]
    dev, T = "cuda:0", torch.float16
    for abl, skew_us in [(0, 0)]:
      print(f"-- ablation {abl} (1 = no epilogue stores, 2 = no residual loads), start skew {skew_us} us")
      lib.lwdetr_gemm_pt_tuning(2 + 256 * ((abl << 24) + skew_us * 100))
      for M, n, k, act, res in [(58368, 2304, 768, K.ACT_NONE, 0), (58368, 768, 768, K.ACT_NONE, 3), (58368, 768, 768, K.ACT_NONE, 1), (58368, 768, 768, K.ACT_NONE, 2),
                                (58368, 768, 768, K.ACT_NONE, 0), (58368, 3072, 768, K.ACT_GELU, 0), (58368, 768, 3072, K.ACT_NONE, 3)]:
          x = torch.randn(M, k, device=dev).to(T)
          w = (torch.randn(n, k, device=dev) * k ** -0.5).to(T)
          b = torch.randn(n, device=dev)
          out = torch.empty(M, n, device=dev, dtype=T)
          r = torch.randn(M, n, device=dev).to(T) if res & 1 else None
          g = torch.rand(n, device=dev) if res & 2 else None
          op = K.GemmOp(x, w, M, n, k, [K.seg(out, 0, n, ldo=n, bias=b, act=act, res=r, ldres=n, gamma=g)])
          for _ in range(3):
              op()
          torch.cuda.synchronize()
          buf = (C.c_ulonglong * 64)()
          assert lib.lwdetr_debug_pt_timing(buf) == 0
          t = [list(buf[8 * wv:8 * wv + 8]) for wv in range(8)]
          tiles = max(t[0][3], 1)
          row = f"M={M} N={n} K={k} res={res & 1} gamma={res >> 1}: workgroup {WG}: {tiles} tiles, {10 * t[0][4] / 1e3:.1f} us in all; per tile [us]"
          row += "  k-loop " + "/".join(f"{10 * t[wv][0] / tiles / 1e3:.2f}" for wv in range(8))
          row += "  epilogue " + "/".join(f"{10 * t[wv][1] / tiles / 1e3:.2f}" for wv in range(8))
          row += "  (first stage wait after an epilogue " + "/".join(f"{10 * t[wv][2] / tiles / 1e3:.2f}" for wv in range(8)) + ", inside the k-loop figure)"
          print(row, flush=True)


if __name__ == "__main__":
    main()
