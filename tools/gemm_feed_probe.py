"""Large-tile GEMM feed probe (tuning): time the kb64 kernel of the library LWDETR_HIP_LIB points at on one shape, with the
real A (lda = K) and with every A row aliased to row 0 (lda = 0: A is L2-resident, only the access pattern remains)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd import _native, kernels as K  # noqa: E402
from tools.gemm_big_bench import timeit  # noqa: E402


def main():
    dev, T = "cuda:0", torch.float16
    lib = _native.lib()
    for M, n, k in [(58368, 768, 3072), (58368, 2304, 768), (51200, 384, 1536)]:
        x = torch.randn(M, k, device=dev).to(T)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(T)
        out = torch.empty(M, n, device=dev, dtype=T)
        row = f"M={M} N={n} K={k}:"
        for mode in (64, 32):
            lib.lwdetr_gemm_tuning(mode)
            for lda in (k, 0):
                op = K.GemmOp(x, w, M, n, k, [K.seg(out, 0, n, ldo=n)], lda=lda)
                us = timeit(op)
                steps = ((M + 255) // 256) * ((n + 255) // 256) * (k // 64) / 256.0
                row += f"  kb{mode} lda={lda:4d} {us:7.1f} us ({us / steps * 2400:6.0f} clk/k64-step)"
        print(row, flush=True)


if __name__ == "__main__":
    main()
