"""The patch-resident 3x3 convolution beside other kernels (debug tool): one convolution problem is repeated on a side stream while
another stream runs a load; every result is compared with the result on an idle GPU and the differing elements are located."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lwdetr_amd import kernels as K


def problem(b, hp, wp, c, seed, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(b * hp * wp, 5 * c, generator=g) * 0.7).to(dtype).cuda()
    w = (torch.randn(c, 9 * c, generator=g) * (9 * c) ** -0.5).to(dtype).cuda()
    bias = torch.randn(c, generator=g).cuda()
    out = torch.zeros(b * hp * wp, c, dtype=dtype, device="cuda")
    op = K.GemmOp(x, w, b * hp * wp, c, 9 * c, [K.seg(out, 0, c, ldo=c, bias=bias, act=K.ACT_SILU)], lda=5 * c, a_mode=K.A_CONV3x3,
                  a_tok=K.tok_layout(False, hp, wp, 0), conv_cin=c, conv_stride=1, a_col0=2 * c, conv_hout=hp, conv_wout=wp, keep=(out, bias))
    return op, out, x


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    c = int(os.environ.get("CONV_C", "128"))
    vop, vout, _ = problem(16, 40, 40, c, 1)
    aop, aout, ax = problem(16, 40, 40, c, 2)
    big = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    pa = torch.randn(25600, 192, device="cuda", dtype=torch.float16) * 0.1; pw = torch.randn(576, 192, device="cuda", dtype=torch.float16) * 0.1
    pout = torch.zeros(25600, 576, device="cuda", dtype=torch.float16)
    gop = K.GemmOp(pa, pw, 25600, 576, 192, [K.seg(pout, 0, 576, ldo=576)])
    el = torch.randn(32 << 20, device="cuda", dtype=torch.float16)
    side = torch.cuda.Stream()
    st0 = torch.cuda.current_stream().cuda_stream
    with torch.cuda.stream(side):
        vop(side.cuda_stream); ref = vout.clone()
        vop(side.cuda_stream); again = vout.clone()
    torch.cuda.synchronize()
    print("idle GPU, repeat:", "identical" if torch.equal(ref, again) else "DIFFERS", flush=True)
    loads = {"none": lambda: None, "the same convolution kernel (another problem) x6": lambda: [aop(st0) for _ in range(6)],
             "plain GEMM 25600 x 576 x 192 (64 x 64 ring kernel) x12": lambda: [gop(st0) for _ in range(12)],
             "torch matmul 4096^3 x2": lambda: [big @ big for _ in range(2)], "torch elementwise 64 MB x6": lambda: [el.mul_(1.0) for _ in range(6)]}
    for name, load in loads.items():
        outs = []
        for r in range(reps):
            load()
            with torch.cuda.stream(side):
                for _ in range(3):
                    vout.fill_(9.0)
                    vop(side.cuda_stream)
                    outs.append(vout.clone())
        torch.cuda.synchronize()
        nbad, shown = 0, 0
        for t, o in enumerate(outs):
            d = o != ref
            if bool(d.any()):
                nbad += 1
                if shown < 5:
                    shown += 1
                    rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
                    tiles = sorted(set((rows // 128).tolist()))
                    eq9 = float((o[d] == 9.0).float().mean())
                    print(f"   trial {t}: {int(d.sum())} elements, {rows.numel()} rows in tiles {tiles[:6]} (row in tile {int(rows.min()) % 128}..{int(rows.max()) % 128}), "
                          f"cols {int(cols.min())}..{int(cols.max())} ({cols.numel()}); fill value left: {eq9:.2f}; max |diff| {float((o.float() - ref.float()).abs().max()):.3f}", flush=True)
        print(f"load {name}: {nbad} of {len(outs)} results differ", flush=True)


main()
