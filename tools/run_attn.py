import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd import kernels as K
T = torch.float16; dev = "cuda:0"; B, heads, hd, tp, C = 32, 12, 16, 1600, 192
q = torch.randn(B, heads, tp, hd, device=dev).to(T); k = torch.randn(B, heads, tp, hd, device=dev).to(T)
vt = torch.randn(B, heads, hd, tp, device=dev).to(T); o = torch.empty(B * tp, C, device=dev, dtype=T)
op = K.AttnOp(q, k, vt, o, B=B, heads=heads, hd=hd, Tp=tp, ldo=C, seqs_per_img=1, seq_tok_stride=tp, keys_per_seq=tp,
              sub_stride=100, sub_len=100, kind=1)
for _ in range(5):
    op()
torch.cuda.synchronize()
