"""GPU: run the code objects of tools/msda_isa_variants.py (the sampling kernel `msda_fused_kernel<f16,1,2>` from the compiler's own
assembly, patched per variant) beside the OTHER launch chain's attention / GEMM kernels and count the launches whose output differs
from the variant's result on an idle GPU.

    python tools/msda_isa_probe.py [reps]      # needs tools/_msda_isa/*.hsaco (built on the CPU box; they travel with gpurun)

The launch is the model's own: chain 1's first decoder layer of LW-DETR-small at B = 32 (16-image parts), same buffers, same grid; the
load is chain 0's AttnOp / GemmOp launches on the other stream, exactly as tools/determinism_probe.py PROBE_MSDA=variants did in round 3."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import lwdetr_amd  # noqa: E402
from lwdetr_amd.models import lwdetr as L  # noqa: E402
from lwdetr_amd.synth import synth_images, synth_state_dict  # noqa: E402

ISA = os.path.join(ROOT, "tools", "_msda_isa")


class MsdaParams(ctypes.Structure):          # csrc/msda.hip:MsdaParams (kernarg, 128 bytes)
    _fields_ = [("value", ctypes.c_void_p), ("shapes", ctypes.c_void_p), ("lsi", ctypes.c_void_p), ("loc", ctypes.c_void_p),
                ("aw", ctypes.c_void_p), ("out", ctypes.c_void_p), ("B", ctypes.c_int), ("S", ctypes.c_int), ("M", ctypes.c_int),
                ("D", ctypes.c_int), ("L", ctypes.c_int), ("Q", ctypes.c_int), ("P", ctypes.c_int), ("chunks_per_img", ctypes.c_int),
                ("xcd_remap", ctypes.c_int), ("oa", ctypes.c_void_p), ("ld_oa", ctypes.c_long), ("ref", ctypes.c_void_p),
                ("vr", ctypes.c_void_p), ("oa_logit_off", ctypes.c_int)]


class Hip:
    def __init__(self):
        self.lib = ctypes.CDLL("libamdhip64.so")

    def check(self, rc, what):
        if rc:
            raise RuntimeError(f"{what}: hip error {rc}")

    def load(self, path, name):
        mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
        self.check(self.lib.hipModuleLoad(ctypes.byref(mod), path.encode()), "hipModuleLoad " + path)
        self.check(self.lib.hipModuleGetFunction(ctypes.byref(fn), mod, name.encode()), "hipModuleGetFunction")
        return fn

    def launch(self, fn, grid, block, params, stream):
        size = ctypes.c_size_t(ctypes.sizeof(params))
        extra = (ctypes.c_void_p * 5)(1, ctypes.cast(ctypes.byref(params), ctypes.c_void_p), 2, ctypes.cast(ctypes.byref(size), ctypes.c_void_p), 3)
        self.check(self.lib.hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, ctypes.c_void_p(stream), None, extra), "hipModuleLaunchKernel")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    meta = json.load(open(os.path.join(ISA, "variants.json")))
    m, _, _ = lwdetr_amd.build_model(lwdetr_amd.get_args("small"))
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0))
    m = m.cuda().half().eval()
    x = synth_images(32, 640, 640, seed=99).cuda().half()
    part = 16
    L.set_streams(2)
    m(x); torch.cuda.synchronize()
    p0, p1 = m._plans[(part, 640, 640, 0)], m._plans[(part, 640, 640, 1)]
    cfg = p1.cfg
    M, D, Lv, P, Q, S, B = cfg.ca_nheads, p1.d // cfg.ca_nheads, p1.L, cfg.dec_n_points, p1.nq, p1.S, part
    assert (Lv, P) == (1, 2) and D % 8 == 0
    bufs = p1.dec_bufs
    ca, oa = bufs["ca"], bufs["oa"]
    prm = MsdaParams(value=p1.values[0].data_ptr(), shapes=p1.shapes_t.data_ptr(), lsi=p1.lsi_t.data_ptr(), loc=None, aw=None,
                     out=ca.data_ptr(), B=B, S=S, M=M, D=D, L=Lv, Q=Q, P=P, chunks_per_img=(Q * M * (D // 8) + 255) // 256,
                     xcd_remap=1 if B % 8 == 0 else 0, oa=oa.data_ptr(), ld_oa=oa.stride(0), ref=p1.ref.data_ptr(), vr=p1.vr.data_ptr(),
                     oa_logit_off=M * Lv * P * 2)
    grid = B * prm.chunks_per_img
    hip = Hip()
    side = torch.cuda.Stream()
    st0 = torch.cuda.current_stream().cuda_stream
    by_type = {}
    for o in list(p0.ops_backbone) + list(p0.ops_enc) + list(p0.ops_dec):
        by_type.setdefault(type(o).__name__, []).append(o)
    # the library's own launch (shipped build) as the reference result
    op = [o for o in p1.ops_dec if type(o).__name__ == "MsdaFusedOp"][0]
    with torch.cuda.stream(side):
        ca.fill_(7.0)
        op(side.cuda_stream)
        ref_lib = ca.clone()
    torch.cuda.synchronize()
    results = {}
    for name, info in meta["variants"].items():
        if only and name not in only:
            continue
        fn = hip.load(os.path.join(ISA, name + ".hsaco"), meta["kernel"])
        with torch.cuda.stream(side):
            ca.fill_(7.0)
            hip.launch(fn, grid, 256, prm, side.cuda_stream)
            ref = ca.clone()
            ca.fill_(7.0)
            hip.launch(fn, grid, 256, prm, side.cuda_stream)
            solo_same = bool(torch.equal(ca, ref))
        torch.cuda.synchronize()
        row = {"equals_shipped_kernel": bool(torch.equal(ref, ref_lib)), "repeats_alone": solo_same, "pk_left": info["pk_left"]}
        for load in ("AttnOp", "GemmOp"):
            bad = torch.zeros(1, dtype=torch.int32, device="cuda")
            lanes = torch.zeros(4, dtype=torch.int64, device="cuda")           # differing elements by lane quarter of the wave
            for r in range(reps):
                for o in by_type[load]:
                    o(st0)
                with torch.cuda.stream(side):
                    for _ in range(4):
                        ca.fill_(7.0)
                        hip.launch(fn, grid, 256, prm, side.cuda_stream)
                        d = ca != ref                                           # (B*Q, 256): a wave = 2 rows; lane = (row & 1) * 32 + col // 8
                        bad += d.any().to(torch.int32)
                        dq = d.view(-1, 2, 2, 16, 8).any(-1).sum((0, 3))        # [row parity][half row] -> lane quarters 0..3
                        lanes += dq.flatten()
            torch.cuda.synchronize()
            row[load] = {"bad": int(bad.item()), "of": 4 * reps, "wrong_lanegroups_by_quarter": lanes.tolist()}
        results[name] = row
        print(f"{name:26s} pk {info['pk_left']:2d}  = shipped: {row['equals_shipped_kernel']!s:5s} repeats alone: {solo_same!s:5s}"
              f"  beside AttnOp: {row['AttnOp']['bad']:3d}/{4 * reps}  beside GemmOp: {row['GemmOp']['bad']:3d}/{4 * reps}"
              f"  lane quarters {row['AttnOp']['wrong_lanegroups_by_quarter']}  | {info['desc']}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "msda_isa_probe.json"), "w"), indent=1)


main()
