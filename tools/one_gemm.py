"""Runs ONE GEMM shape a few times (profiling target for rocprofv3 --pmc):  python tools/one_gemm.py M N K [mode] [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd import _native, kernels as K  # noqa: E402

M, N, Kk = (int(v) for v in sys.argv[1:4])
mode = int(sys.argv[4]) if len(sys.argv) > 4 else -1
T = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[5] if len(sys.argv) > 5 else "fp16"]
dev = "cuda:0"
x = torch.randn(M, Kk, device=dev).to(T)
w = (torch.randn(N, Kk, device=dev) * Kk ** -0.5).to(T)
b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=T)
op = K.GemmOp(x, w, M, N, Kk, [K.seg(out, 0, N, ldo=N, bias=b)])
_native.lib().lwdetr_gemm_tuning(mode)
for _ in range(6):
    op()
torch.cuda.synchronize()
