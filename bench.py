"""Benchmark of the LW-DETR inference hot path on MI355X (contract: see the task's bench.py section).

    python bench.py [--gpus N --steps K --warmup W]          # N=1: LW-DETR-small, 640x640, batch 32, fp16
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one forward of the HIP path over one resident synthetic batch (images already in HBM) + PostProcess
(+ the RCCL all-gather of detections when N > 1; weak scaling: the per-GPU batch is fixed). Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import lwdetr_amd  # noqa: E402
from lwdetr_amd import _native, dist as ldist  # noqa: E402
from lwdetr_amd.configs import GFLOP_PER_IMAGE  # noqa: E402
from lwdetr_amd.synth import synth_images, synth_state_dict  # noqa: E402

PEAK_TFLOPS = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3}       # dense MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", default="small", choices=lwdetr_amd.SIZES)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--res", type=int, default=640)
    ap.add_argument("--dtype", default="fp16", choices=list(DTYPES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--latency", action="store_true", help="also report p50/p90 single-image latency (bs=1)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def _cpu_baseline_worker(size, res):
    from oracle import lwdetr_torch as O
    cores = torch.get_num_threads()                 # torch's default: the cores this process may actually use
    cfg = lwdetr_amd.get_args(size)
    model, _, _ = lwdetr_amd.build_model(cfg)
    sd = synth_state_dict(model.state_dict(), seed=0)
    b = 4
    x = synth_images(b, res, res, seed=1234)
    with torch.no_grad():
        O.forward(sd, cfg, x)                       # warm-up
        t0, n = time.time(), 0
        while n < 3 or (time.time() - t0 < 10.0 and n < 12):
            O.forward(sd, cfg, x)
            n += 1
        dt = time.time() - t0
    print(json.dumps({"value": round(b * n / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
                      "sample": f"{n} forwards of batch {b} at {res}x{res}, fp32, oracle/lwdetr_torch.py "
                                f"(CPU restatement of the reference PyTorch path) on {cores} threads"}))


def cpu_baseline(size, res):
    """The CPU oracle (a port of the reference's PyTorch CPU path, pinned to reference goldens) timed on this box's
    host cores on a bounded sample of the same workload; runs in a child process under a hard time limit."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--size", size,
                            "--res", str(res)], capture_output=True, text=True, timeout=150)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001 - the baseline is informational; never lose the GPU measurement
        return {"value": None, "unit": "images/sec", "cores": None, "kind": "port", "sample": f"failed: {e!r}"[:200]}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    if a.cpu_baseline_worker:
        return _cpu_baseline_worker(a.size, a.res)
    rank, world, local = ldist.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a ROCm device is required"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    _native.lib()
    T = DTYPES[a.dtype]
    cfg = lwdetr_amd.get_args(a.size)
    model, _, post = lwdetr_amd.build_model(cfg)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).to(T).eval()
    pp = post["bbox"]
    images = synth_images(a.batch, a.res, a.res, seed=1234 + rank).to(dev).to(T)
    sizes = torch.tensor([[480.0, 640.0]] * a.batch, device=dev)
    gathered = torch.empty(world * a.batch, cfg.num_select, 6, dtype=torch.float32, device=dev) if world > 1 else None

    def step():
        out = model(images)
        s, l, b = pp.select(out["pred_logits"], out["pred_boxes"], sizes)
        det = ldist.pack_detections(s, l, b)
        return ldist.all_gather_detections(det, gathered)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    log(f"model built ({a.size}, {a.dtype}); warm-up x{a.warmup}")
    for _ in range(a.warmup):
        step()
    barrier()
    log("timing")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        det = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(det).all()
    log(f"timed {a.steps} steps: {dt / a.steps * 1e3:.3f} ms/step")
    ms_step = dt / a.steps * 1e3
    ips = world * a.batch * a.steps / dt

    result = {
        "metric": "images/sec", "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"LW-DETR-{a.size} inference forward + PostProcess, {a.res}x{a.res}, batch {a.batch}/GPU, "
                               f"{a.dtype}, random-init weights (synthetic COCO-shaped input)",
                   "global_batch": world * a.batch, "per_gpu_batch": a.batch, "parallelism": f"dp{world}",
                   "collective": "all_gather_into_tensor of (B,K,6) f32 detections" if world > 1 else "none"},
    }
    gf = GFLOP_PER_IMAGE.get((a.size, a.res))
    if gf:
        result["model_tflops"] = round(ips * gf / 1e3, 2)
        result["model_mfma_frac"] = round(ips * gf / 1e3 / (PEAK_TFLOPS[a.dtype] * world), 4)

    if rank == 0 and not a.no_roofline:
        # dedicated pass with per-kernel HIP events on the launch stream (outside the timed region)
        _native.prof_enable(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        prof = _native.prof_collect()
        _native.prof_enable(False)
        tot = sum(v["ms"] for v in prof.values())
        table = {k: {"ms_per_step": round(v["ms"] / 3, 4), "launches_per_step": v["count"] // 3,
                     "share": round(v["ms"] / tot, 4)} for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        name, v = max(prof.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = v["ms"] / v["count"]
        fl, by = v["flops"] / v["count"], v["bytes"] / v["count"]
        ridge = PEAK_TFLOPS[a.dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
        if fl > 0 and fl / max(by, 1.0) >= ridge:
            ach = fl / (avg_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[a.dtype], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS[a.dtype], 4)}
        else:
            ach = by / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4)}
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")           # tools/profile_round.sh, this workload
        if os.path.exists(tf) and (a.size, a.batch, a.res, a.dtype) == ("small", 32, 640, "fp16"):
            traffic = json.load(open(tf)).get(name, {}).get("bytes_per_launch")      # rocprofv3 PMC pass of this workload
        roof.update({"kernel": name, "avg_launch_us": round(avg_ms * 1e3, 2), "launches": v["count"],
                     "alg_flops_per_launch": fl, "alg_bytes_per_launch": by, "traffic": traffic})
        result["roofline"] = roof
        result["kernels"] = table

    if rank == 0 and a.latency:
        one = images[:1].contiguous()

        def p50p90(fn):
            lat = []
            for i in range(110):
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                fn()
                torch.cuda.synchronize(dev)
                if i >= 10:
                    lat.append((time.perf_counter() - t) * 1e3)
            lat.sort()
            return {"p50": round(lat[len(lat) // 2], 3), "p90": round(lat[int(len(lat) * 0.9)], 3)}

        # forward + PostProcess of one resident image: eager launches, then the same work replayed as one HIP graph
        result["latency_bs1_ms"] = p50p90(lambda: pp.select(*(lambda o: (o["pred_logits"], o["pred_boxes"]))(model(one)), sizes[:1]))
        try:
            graphed = model.capture(one, postprocess=pp, target_sizes=sizes[:1])
            result["latency_bs1_hipgraph_ms"] = p50p90(lambda: graphed(one))
        except Exception as e:                                   # report, never hide: the eager number above stands
            result["latency_bs1_hipgraph_ms"] = {"error": repr(e)[:200]}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("cpu baseline (child process)")
        result["cpu_baseline"] = cpu_baseline(a.size, a.res)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()          # rank 0 may still be in its (untimed) roofline pass: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
