"""Which stage of the forward is not bit-reproducible? (debug tool)
  python tools/determinism_probe.py <size> <batch> <runs> [chains]
chains = 1: the same batch N times on one launch chain, collected tensors compared with the first run.
chains <= -2: as chains >= 2, and every decoder op of every plan is followed by on-stream copies of the decoder's buffers: reports the
first op after which a buffer differs.
chains >= 2: the batch as `chains` launch chains (LWDETR._forward_chains), every run compared - per chain, per internal buffer of its
plan - with the same part run alone on one stream."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lwdetr_amd
from lwdetr_amd.models import lwdetr as L
from lwdetr_amd.synth import synth_images, synth_state_dict

NAMES = ["x", "taps_cat", "memory", "om", "cls_max", "topk_idx", "hs"]


def snap(plan):
    return {k: getattr(plan, k).clone() for k in NAMES if torch.is_tensor(getattr(plan, k, None))}


def diff(a, b):
    out = {}
    for k in a:
        if k in b and a[k].shape == b[k].shape:
            n = int((a[k] != b[k]).sum().item())
            if n:
                rows = (a[k].reshape(a[k].shape[0], -1) != b[k].reshape(b[k].shape[0], -1)).any(1).nonzero().flatten()
                out[k] = (n, float((a[k].float() - b[k].float()).abs().max().item()), rows[:6].tolist(), int(rows.numel()))
    return out


class Traced:
    """Wraps one decoder op of a plan: after the launch, copies of the decoder's buffers on the same stream."""
    def __init__(self, op, plan, idx, log):
        self.op, self.plan, self.idx, self.log = op, plan, idx, log

    def __call__(self, stream):
        self.op(stream)
        self.log.append((self.idx, type(self.op).__name__, {k: v.clone() for k, v in self.plan.dec_bufs.items()}))


def decoder_trace(m, x, b, n, nch, part):
    logs = {}

    def wrap(plan, key):
        if not isinstance(plan.ops_dec[0], Traced):
            logs[key] = []
            plan.ops_dec = [Traced(op, plan, i, logs[key]) for i, op in enumerate(plan.ops_dec)]

    L.set_streams(1)
    m(x[:part]); torch.cuda.synchronize()
    wrap(m._plans[(part, 640, 640, 0)], 0)
    refs = []
    for i in range(nch):
        m(x[i * part:(i + 1) * part])          # twice: buffers an op has not yet written hold the previous forward of the SAME part,
        logs[0].clear()                         # as they do in the repeated chained runs below
        m(x[i * part:(i + 1) * part]); torch.cuda.synchronize()
        refs.append(list(logs[0]))
    L.set_streams(nch)
    logs[0].clear()
    m(x); torch.cuda.synchronize()
    for i in range(1, nch):
        wrap(m._plans[(part, 640, 640, i)], i)
    nbad = 0
    for it in range(n):
        for lg in logs.values():
            lg.clear()
        m(x); torch.cuda.synchronize()
        for i in range(nch):
            for (idx, name, bufs), (_, _, rb) in zip(logs[i], refs[i]):
                bad = diff(bufs, rb)
                if bad:
                    nbad += 1
                    short = {k: (v[0], round(v[1], 4), v[2][:4], v[3]) for k, v in bad.items()}
                    if nbad <= 8:
                        print(f"run {it} chain {i}: first difference after decoder op {idx} ({name}): {short}", flush=True)
                    break
    print(f"{nch} chains, {n} runs, decoder traced: {nbad} chain-runs differ", flush=True)


def op_stress(m, x, part, reps):
    """Every decoder op of chain 1's plan alone on a side stream, repeated from the same buffer state, while the other part's whole
    forward runs on the current stream: which op gives different results under load?"""
    L.set_streams(2)
    m(x); torch.cuda.synchronize()
    p0, p1 = m._plans[(part, 640, 640, 0)], m._plans[(part, 640, 640, 1)]
    bufs = dict(p1.dec_bufs, logits=p1.logits, delta=p1.delta)
    saved = {k: v.clone() for k, v in bufs.items()}
    side = torch.cuda.Stream()
    names = list(bufs)

    def restore():
        for k in names:
            bufs[k].copy_(saved[k])

    for j, op in enumerate(p1.ops_dec):
        with torch.cuda.stream(side):
            restore(); op(side.cuda_stream)
            ref = {k: bufs[k].clone() for k in names}
            restore(); op(side.cuda_stream)
            solo = sum(int((bufs[k] != ref[k]).any().item()) for k in names)
            flags = torch.zeros(len(names), dtype=torch.int32, device=x.device)
        torch.cuda.synchronize()
        for r in range(reps):
            p0.run(x[:part])
            with torch.cuda.stream(side):
                for _ in range(4):
                    restore(); op(side.cuda_stream)
                    for idx, k in enumerate(names):
                        flags[idx] += (bufs[k] != ref[k]).any().to(torch.int32)
        torch.cuda.synchronize()
        f = flags.tolist()
        bad = {k: f[i] for i, k in enumerate(names) if f[i]}
        print(f"op {j:2d} {type(op).__name__:16s} solo-repeat-differs={solo} under-load: {bad or 'identical'} of {4 * reps}", flush=True)


def msda_stress(m, x, part, reps):
    """The decoder's sampling ops under load, with every trial's output kept: where do the differing elements sit, and what do they hold?"""
    L.set_streams(2)
    m(x); torch.cuda.synchronize()
    p0, p1 = m._plans[(part, 640, 640, 0)], m._plans[(part, 640, 640, 1)]
    bufs = dict(p1.dec_bufs)
    saved = {k: v.clone() for k, v in bufs.items()}
    side = torch.cuda.Stream()
    mode = os.environ.get("PROBE_STRESS", "1")
    for j, op in enumerate(p1.ops_dec):
        if type(op).__name__ != "MsdaFusedOp":
            continue
        with torch.cuda.stream(side):
            for k in bufs:
                bufs[k].copy_(saved[k])
            op(side.cuda_stream)
            ref = bufs["ca"].clone()
        torch.cuda.synchronize()
        outs = []
        for r in range(reps):
            p0.run(x[:part])
            with torch.cuda.stream(side):
                for _ in range(4):
                    if mode == "2":
                        bufs["ca"].fill_(7.0)              # a value neither the saved nor the new output holds
                    else:
                        bufs["ca"].copy_(saved["ca"])
                    op(side.cuda_stream)
                    outs.append(bufs["ca"].clone())
        torch.cuda.synchronize()
        nb = 0
        for t, o in enumerate(outs):
            d = o != ref
            if bool(d.any()):
                nb += 1
                rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
                eq_saved = float((o[d] == saved["ca"][d]).float().mean()); eq7 = float((o[d] == 7.0).float().mean())
                print(f"op {j} trial {t}: {int(d.sum())} elements differ; rows {rows[:8].tolist()} ({rows.numel()}), cols {cols.min().item()}..{cols.max().item()} ({cols.numel()});"
                      f" equal to the restored value {eq_saved:.2f}, to the fill value {eq7:.2f}; max |diff| {float((o.float() - ref.float()).abs().max()):.3f}", flush=True)
        print(f"op {j}: {nb} of {len(outs)} trials differ", flush=True)


def msda_loads(m, x, part, reps):
    """The first sampling op of chain 1's plan under different co-running loads; its inputs are checked for changes as well."""
    L.set_streams(2)
    m(x); torch.cuda.synchronize()
    p0, p1 = m._plans[(part, 640, 640, 0)], m._plans[(part, 640, 640, 1)]
    bufs = dict(p1.dec_bufs)
    saved = {k: v.clone() for k, v in bufs.items()}
    side = torch.cuda.Stream()
    ops = [op for op in p1.ops_dec if type(op).__name__ == "MsdaFusedOp"]
    op = ops[0]
    inputs = dict(values=p1.values[0], ref=p1.ref, vr=p1.vr, oa=bufs["oa"])
    in_saved = {k: v.clone() for k, v in inputs.items()}
    a = torch.randn(4096, 4096, device=x.device, dtype=torch.float16); e = torch.randn(64 << 20, device=x.device, dtype=torch.float16)
    st0 = torch.cuda.current_stream().cuda_stream
    by_type = {}
    for o in list(p0.ops_backbone) + list(p0.ops_enc) + list(p0.ops_dec):
        by_type.setdefault(type(o).__name__, []).append(o)

    def run_ops(lst, k=1):
        def f():
            for _ in range(k):
                for o in lst:
                    o(st0)
        return f

    loads = {"none": lambda: None, "whole forward of part 0": lambda: p0.run(x[:part]), "backbone ops": run_ops(p0.ops_backbone),
             "torch matmul 4096^3 x8": lambda: [a @ a for _ in range(8)], "torch elementwise 128 MB x8": lambda: [e.mul_(1.0) for _ in range(8)]}
    for name, lst in by_type.items():
        loads[f"{name} x{len(lst)} of part 0"] = run_ops(lst, 3 if len(lst) < 20 else 1)
    with torch.cuda.stream(side):
        for k in bufs:
            bufs[k].copy_(saved[k])
        op(side.cuda_stream)
        ref = bufs["ca"].clone()
    torch.cuda.synchronize()
    for name, load in loads.items():
        bad = torch.zeros(1, dtype=torch.int32, device=x.device); bad_in = torch.zeros(1, dtype=torch.int32, device=x.device)
        for r in range(reps):
            load()
            with torch.cuda.stream(side):
                for _ in range(4):
                    bufs["ca"].fill_(7.0)
                    op(side.cuda_stream)
                    bad += (bufs["ca"] != ref).any().to(torch.int32)
                    for k, v in inputs.items():
                        bad_in += (v != in_saved[k]).any().to(torch.int32)
        torch.cuda.synchronize()
        print(f"load {name:40s}: {int(bad.item()):3d} of {4 * reps} trials differ; input buffers changed in {int(bad_in.item())} checks", flush=True)


def msda_variants(m, x, part, reps):
    """Experimental forms of the sampling kernel (LWDETR_MSDA_VAR) under the two loads that disturb it."""
    L.set_streams(2)
    m(x); torch.cuda.synchronize()
    p0, p1 = m._plans[(part, 640, 640, 0)], m._plans[(part, 640, 640, 1)]
    bufs = dict(p1.dec_bufs)
    saved = {k: v.clone() for k, v in bufs.items()}
    side = torch.cuda.Stream()
    op = [o for o in p1.ops_dec if type(o).__name__ == "MsdaFusedOp"][0]
    st0 = torch.cuda.current_stream().cuda_stream
    by_type = {}
    for o in list(p0.ops_backbone) + list(p0.ops_enc) + list(p0.ops_dec):
        by_type.setdefault(type(o).__name__, []).append(o)
    ref0 = None
    for var in [int(v) for v in os.environ.get("PROBE_VARS", "0,1,3,4,5,6").split(",")]:
        os.environ["LWDETR_MSDA_VAR"] = str(var)
        with torch.cuda.stream(side):
            for k in bufs:
                bufs[k].copy_(saved[k])
            op(side.cuda_stream)
            ref = bufs["ca"].clone()
        torch.cuda.synchronize()
        if ref0 is None:
            ref0 = ref
        line = f"variant {var}: solo result equals variant 0: {bool(torch.equal(ref, ref0))};"
        for name in ("AttnOp", "GemmOp"):
            bad = torch.zeros(1, dtype=torch.int32, device=x.device)
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            for r in range(reps):
                for o in by_type[name]:
                    o(st0)
                with torch.cuda.stream(side):
                    for _ in range(4):
                        bufs["ca"].fill_(7.0)
                        op(side.cuda_stream)
                        bad += (bufs["ca"] != ref).any().to(torch.int32)
            torch.cuda.synchronize()
            line += f" beside {name}: {int(bad.item())} of {4 * reps} differ;"
        print(line, flush=True)
    os.environ.pop("LWDETR_MSDA_VAR", None)


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "small"
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    nch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    m, _, _ = lwdetr_amd.build_model(lwdetr_amd.get_args(size))
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0)); m = m.cuda().half().eval()
    x = synth_images(b, 640, 640, seed=99).cuda().half()
    if nch == 1:
        ref = None
        for it in range(n):
            col = {}
            out = m(x, _collect=col)
            torch.cuda.synchronize()
            cur = {k: v.clone() for k, v in col.items() if torch.is_tensor(v)}
            cur["pred_logits"] = out["pred_logits"].clone()
            if ref is None:
                ref = cur; continue
            bad = diff(cur, ref)
            print(f"run {it}: " + ("identical" if not bad else str(bad)), flush=True)
        return
    trace = nch < 0
    nch = abs(nch)
    part = b // nch
    if trace and os.environ.get("PROBE_MSDA") == "variants":
        return msda_variants(m, x, part, n)
    if trace and os.environ.get("PROBE_MSDA") == "loads":
        return msda_loads(m, x, part, n)
    if trace and os.environ.get("PROBE_STRESS") in ("1", "2") and os.environ.get("PROBE_MSDA"):
        return msda_stress(m, x, part, n)
    if trace and os.environ.get("PROBE_STRESS"):
        return op_stress(m, x, part, n)
    if trace:
        return decoder_trace(m, x, b, n, nch, part)
    L.set_streams(1)
    refs = []
    for i in range(nch):                       # every part alone on one stream, through plan slot 0
        o = m(x[i * part:(i + 1) * part])
        torch.cuda.synchronize()
        s = snap(m._plans[(part, 640, 640, 0)]); s["pred_logits"] = o["pred_logits"].clone(); s["pred_boxes"] = o["pred_boxes"].clone()
        refs.append(s)
    o = m(x[:part]); torch.cuda.synchronize()
    s = snap(m._plans[(part, 640, 640, 0)]); s["pred_logits"] = o["pred_logits"].clone()
    print("one chain, part 0 repeated:", diff(s, refs[0]) or "identical", flush=True)
    L.set_streams(nch)
    nbad = 0
    for it in range(n):
        o = m(x)
        torch.cuda.synchronize()
        for i in range(nch):
            s = snap(m._plans[(part, 640, 640, i)])
            s["pred_logits"] = o["pred_logits"][i * part:(i + 1) * part]; s["pred_boxes"] = o["pred_boxes"][i * part:(i + 1) * part]
            bad = diff(s, refs[i])
            if bad:
                nbad += 1
                print(f"run {it} chain {i}: {bad}", flush=True)
    print(f"{nch} chains, {n} runs: {nbad} chain-runs differ from the part run alone", flush=True)


main()
