"""Per-launch times of one forward (tuning tool): every op of the launch plan timed with HIP events, 20 repetitions."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def describe(op, K):
    n = type(op).__name__
    if isinstance(op, K.GemmOp):
        d = op.desc
        segs = [(d.seg[i].n_begin, d.seg[i].n_end, d.seg[i].mode, d.seg[i].act, bool(d.seg[i].res), bool(d.seg[i].gamma)) for i in range(d.nseg)]
        return f"Gemm M={d.M} N={d.N} K={d.K} amode={d.a_mode} a2={bool(d.A2)} segs={segs}"
    if isinstance(op, K.AttnOp):
        d = op.desc
        return f"Attn B={d.B} heads={d.heads} hd={d.hd} keys={d.keys_per_seq} seqs={d.seqs_per_img} kind={d.kind}"
    if isinstance(op, K.RawOp):
        return f"Raw {op.name}"
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="small")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--res", type=int, default=640)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--gemm-big", type=int, default=-1, help="lwdetr_gemm_tuning mode: compare every GEMM with mode 0 / this mode")
    a = ap.parse_args()
    import torch
    import lwdetr_amd
    from lwdetr_amd import _native, kernels as K
    from lwdetr_amd.synth import synth_images, synth_state_dict
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[a.dtype]
    dev = torch.device("cuda:0")
    model, _, _ = lwdetr_amd.build_model(lwdetr_amd.get_args(a.size))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).to(dt).eval()
    x = synth_images(a.batch, a.res, a.res, seed=1).to(dev).to(dt)
    from lwdetr_amd.models import lwdetr as L
    L.set_streams(1)            # one launch chain: the plan timed below is the one the warm-up ran (its image pointer is set by run())
    for _ in range(3):
        model(x)
    plan = model._plan(a.batch, a.res, a.res)
    stream = _native.stream_ptr(dev)
    groups = [("backbone", plan.ops_backbone), ("enc", plan.ops_enc), ("sel", [plan.op_rowmax, plan.op_topk, plan.op_gather] + list(plan.ops_sel) + [plan.op_dec_inputs]),
              ("dec", list(plan.ops_dec) + [plan.op_finalize])]
    total = 0.0
    for gname, ops in groups:
        for i, op in enumerate(ops):
            ts = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                op(stream)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            med = ts[len(ts) // 2]
            total += med
            extra = ""
            if a.gemm_big >= 0 and isinstance(op, K.GemmOp):
                _native.lib().lwdetr_gemm_tuning(a.gemm_big)
                t2 = []
                for _ in range(20):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    op(stream)
                    e1.record()
                    torch.cuda.synchronize()
                    t2.append(e0.elapsed_time(e1) * 1e3)
                _native.lib().lwdetr_gemm_tuning(-1)
                t2.sort()
                extra = f"   [gemm_tuning({a.gemm_big}): {t2[len(t2) // 2]:8.1f} us]"
            print(f"{gname:8s} {i:3d} {med:8.1f} us  {describe(op, K)}{extra}")
    print(f"sum of medians: {total:.1f} us")


if __name__ == "__main__":
    main()
