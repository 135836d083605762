import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd import _native, kernels as K
lib = _native.lib()
dev = "cuda:0"
torch.manual_seed(0)
M, N, Kd = 1024, 768, 256
T = torch.float16
x = torch.randn(M, Kd, device=dev).to(T)
w = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(T)
lib.lwdetr_gemm_tuning(2)
outs = {}
for mode in (2, 0):
    lib.lwdetr_gemm_pt_tuning(mode)
    out = torch.zeros(M, N, device=dev, dtype=T)
    K.GemmOp(x, w, M, N, Kd, [K.seg(out, 0, N, ldo=N)])()
    torch.cuda.synchronize()
    outs[mode] = out.float()
print("pt launches", lib.lwdetr_gemm_pt_count())
ref = x.float() @ w.float().t()
print("big vs ref", (outs[0] - ref).abs().max().item(), "pt vs ref", (outs[2] - ref).abs().max().item())
err = (outs[2] - ref).abs()
# per 32x32 block error map of the first tile
blk = err[:256, :256].reshape(8, 32, 8, 32).amax(dim=(1, 3))
print("tile (0,0) 32x32-block max err:\n", blk)
# rows / cols wrong inside first 32x32 block
print("first block err by row:", err[:32, :32].amax(dim=1))
print("first block err by col:", err[:32, :32].amax(dim=0))
# does pt output match ref with some permutation? check whether out row r equals ref row r' for some r'
o = outs[2]
for r in range(0, 4):
    d = (ref[:256, :32] - o[r, :32]).abs().amax(dim=1)
    print("pt row", r, "closest ref row", d.argmin().item(), d.min().item())
for c in range(0, 4):
    d = (ref[:32, :256] - o[:32, c:c + 1]).abs().amax(dim=0)
    print("pt col", c, "closest ref col", d.argmin().item(), d.min().item())
# k-partial check: is pt = partial sums? compare with ref computed from k subsets
for k0 in range(0, Kd, 64):
    part = x[:, k0:k0 + 64].float() @ w[:, k0:k0 + 64].float().t()
    print("stage", k0 // 64, "corr with (pt - ref):", torch.corrcoef(torch.stack([(o - ref)[:256, :256].flatten(), part[:256, :256].flatten()]))[0, 1].item())
