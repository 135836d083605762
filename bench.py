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

# dmabuf IPC: the ROCm host driver of this pool does not support the legacy IPC mode; RCCL's peer-to-peer setup over xGMI
# (hipIpcGetMemHandle) needs this BEFORE the HSA runtime starts, i.e. before the first device call (see lwdetr_amd.dist)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

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
    ap.add_argument("--latency", action="store_true", help="also report p50/p90 single-image latency (bs=1); on by default at N=1")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-image latency measurement")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other BASELINE configurations (N=1 default workload only)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-cores", default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help=argparse.SUPPRESS)
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)   # CPU/gloo check of the self-launch
    return ap.parse_args()


def _cpu_forward_fn(size, res, batch):
    """-> (kind, callable running one fp32 CPU forward of `batch` images). The unmodified reference (through the import
    shims of oracle/ref_shims.py) when /root/reference exists - the build container; on the GPU box it does not, and the
    oracle restatement of the same PyTorch path (pinned to reference outputs in tests/) is timed instead."""
    cfg = lwdetr_amd.get_args(size)
    x = synth_images(batch, res, res, seed=1234)
    from oracle import ref_shims
    if ref_shims.reference_available():
        model, _ = ref_shims.build_reference_model(cfg)
        model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
        return "reference", lambda: model(x)
    from oracle import lwdetr_torch as O
    model, _, _ = lwdetr_amd.build_model(cfg)
    sd = synth_state_dict(model.state_dict(), seed=0)
    return "port", lambda: O.forward(sd, cfg, x)


def _cpu_baseline_worker(a):
    """One worker: `--cpu-threads` torch threads pinned to `--cpu-cores`, forwards of batch 2 for ~`--cpu-seconds`."""
    if a.cpu_cores:
        lo, hi = (int(v) for v in a.cpu_cores.split("-"))
        try:
            os.sched_setaffinity(0, set(range(lo, hi + 1)))
        except OSError:
            pass
    if a.cpu_threads:
        torch.set_num_threads(a.cpu_threads)
    b = 2
    kind, fwd = _cpu_forward_fn(a.size, a.res, b)
    with torch.no_grad():
        fwd()                                       # warm-up
        print("READY", flush=True)
        sys.stdin.readline()                        # the parent releases all workers together
        t0, n = time.time(), 0
        while n < 2 or time.time() - t0 < a.cpu_seconds:
            fwd()
            n += 1
        dt = time.time() - t0
    print(json.dumps({"images": b * n, "seconds": dt, "kind": kind, "threads": torch.get_num_threads()}), flush=True)


def _cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _run_cpu_workers(size, res, procs, threads, cores, seconds):
    """`procs` concurrent workers x `threads` torch threads, each pinned to its own slice of `cores` host cores;
    returns aggregate images/sec over the common measurement window."""
    import subprocess
    per = max(1, cores // procs)
    ws = []
    for i in range(procs):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--size", size, "--res", str(res),
               "--cpu-threads", str(threads), "--cpu-cores", f"{i * per}-{i * per + per - 1}", "--cpu-seconds", str(seconds)]
        ws.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    for w in ws:                                     # all warmed up ...
        assert w.stdout.readline().strip() == "READY"
    for w in ws:                                     # ... start together
        w.stdin.write("go\n"); w.stdin.flush()
    res_ = [json.loads(w.stdout.readline()) for w in ws]
    for w in ws:
        w.wait(timeout=60)
    return sum(r["images"] / r["seconds"] for r in res_), res_[0]["kind"]


def cpu_baseline(size, res):
    """The reference's PyTorch CPU path timed on this box's host cores on a bounded sample of the same workload
    (forwards of batch 2, fp32). Images are independent, so the honest use of a many-core host is several worker
    processes; the sweep times 1 x all-cores, and N x 8 / N x 16 threads with every worker pinned to its own cores, and
    reports the best arrangement (all of them are listed in `sweep`)."""
    cores = len(os.sched_getaffinity(0))
    quota = _host_cpu_facts().get("cgroup_quota_cores")
    # Round 6: arrangements that start more threads than twice the container's CPU quota only measure the throttle (the GPU
    # box shows 256 logical CPUs with 16 CPUs' worth of time: 32 x 8 threads ran at load average 27 with 800 throttled periods) - the thread
    # budget of the sweep is min(visible CPUs, 2 x quota); and the winner is run three times in all: `value` = the median, `spread` = min / max.
    budget = cores if not quota else max(2, min(cores, int(round(2 * quota))))
    plans = [(1, min(budget, 32))]
    for t in (8, 16, 4):
        if budget // t >= 2 and (budget // t, t) not in plans:
            plans.append((budget // t, t))
    sweep, best, kind = [], None, "port"
    t_begin = time.time()
    for procs, threads in plans:
        if time.time() - t_begin > 70:
            break
        try:
            ips, kind = _run_cpu_workers(size, res, procs, threads, cores, 5.0)
        except Exception as e:      # noqa: BLE001 - the baseline is informational; never lose the GPU measurement
            sweep.append({"procs": procs, "threads": threads, "error": repr(e)[:120]})
            continue
        sweep.append({"procs": procs, "threads": threads, "images_per_sec": round(ips, 2)})
        if best is None or ips > best[0]:
            best = (ips, procs, threads)
    if best is None:
        return {"value": None, "unit": "images/sec", "cores": cores, "kind": kind, "sample": "failed", "sweep": sweep}
    reruns = [best[0]]
    for _ in range(2):
        if time.time() - t_begin > 130:
            break
        try:
            reruns.append(_run_cpu_workers(size, res, best[1], best[2], cores, 5.0)[0])
        except Exception:       # noqa: BLE001
            break
    reruns.sort()
    best = (reruns[len(reruns) // 2], best[1], best[2])
    what = ("the unmodified reference (oracle/ref_shims.py import shims)" if kind == "reference" else
            "oracle/lwdetr_torch.py (CPU restatement of the reference PyTorch path; /root/reference is absent on this box)")
    host = _host_cpu_facts()
    used = best[1] * best[2]
    quota = host.get("cgroup_quota_cores")
    # `cores` = what the workers could actually run on: the threads started, capped by the cgroup CPU quota of the box's
    # container (the GPU box shows 256 logical CPUs with a quota of 16 CPUs' worth of time: threads beyond it are throttled)
    eff = used if not quota else min(used, max(1, int(round(quota))))
    return {"value": round(best[0], 2), "unit": "images/sec", "cores": eff, "threads_started": used, "kind": kind,
            "runs_of_the_winner": [round(v, 2) for v in reruns], "spread": [round(reruns[0], 2), round(reruns[-1], 2)],
            "cpu_model": _cpu_model_string(), "host_cores": cores, "host": host, "thread_budget": budget,
            "sample": f"{best[1]} worker process(es) x {best[2]} threads, ~5 s of batch-2 forwards each at {res}x{res}, "
                      f"fp32, {what}; value = median of {len(reruns)} runs of this arrangement", "sweep": sweep}


def _host_cpu_facts():
    """What the box actually grants this process: cgroup CPU quota, affinity / online CPUs, load average (the baseline on the
    256-thread GPU host scales far below linearly; these are the facts that explain it, recorded instead of guessed)."""
    facts = {"affinity": len(os.sched_getaffinity(0)), "nproc_online": os.cpu_count()}
    for name, path in (("cgroup_cpu_max", "/sys/fs/cgroup/cpu.max"), ("cgroup_v1_cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
                       ("cgroup_v1_cfs_period_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"), ("cpuset_effective", "/sys/fs/cgroup/cpuset.cpus.effective")):
        try:
            facts[name] = open(path).read().strip()[:80]
        except OSError:
            pass
    try:
        q, per = facts.get("cgroup_cpu_max", "max 100000").split()
        if q != "max":
            facts["cgroup_quota_cores"] = round(int(q) / int(per), 2)
        elif int(facts.get("cgroup_v1_cfs_quota_us", "-1")) > 0:
            facts["cgroup_quota_cores"] = round(int(facts["cgroup_v1_cfs_quota_us"]) / int(facts.get("cgroup_v1_cfs_period_us", "100000")), 2)
    except (ValueError, ZeroDivisionError):
        pass
    try:
        facts["loadavg"] = open("/proc/loadavg").read().split()[:3]
    except OSError:
        pass
    try:
        st = {ln.split()[0]: ln.split()[1:] for ln in open("/sys/fs/cgroup/cpu.stat")}
        facts["cgroup_nr_throttled"] = st.get("nr_throttled", [None])[0]
    except OSError:
        pass
    return facts


def self_launch(a):
    """`python bench.py --gpus N` without a launcher environment: start N ranks on this node (one per GPU) under
    torch.distributed.run - the same command line the driver uses - and pass their exit code on.
    Reference counterpart: the ranks come from the launcher env, util/misc.py:388-439, main.py:206-224."""
    import socket
    import subprocess
    if not a.launch_check and torch.cuda.device_count() < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"self-launch: {' '.join(cmd[1:8])} ...")
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ)))      # HSA_ENABLE_IPC_MODE_LEGACY=0 travels with it


def launch_check(rank, world):
    """CPU / gloo check of the launch path (tests/test_dist_gloo.py): every rank joins the group, contributes its shard
    to the one all-gather of the data path, rank 0 prints the line with the world size it actually saw."""
    det = torch.full((2, 3, 6), float(rank))
    full = ldist.all_gather_detections(det)
    ok = full.shape[0] == 2 * world and all(float(full[2 * r, 0, 0]) == r for r in range(world))
    torch.distributed.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": bool(ok), "n_gpus": world, "backend": torch.distributed.get_backend()}), flush=True)
    torch.distributed.destroy_process_group()


# The other BASELINE configurations as one GPU sees them (BASELINE.json configs 3-5: the per-GPU shard of the 8-GPU ones) + tiny:
# measured after the timed region of the default workload and reported under "other_configs" of the same JSON line.
OTHER_CONFIGS = [("tiny", 32, 640, "fp16"), ("medium", 64, 640, "bf16"), ("large", 32, 640, "fp16"), ("xlarge", 16, 960, "fp16")]


def run_other_config(size, batch, res, dtype, dev, steps=20, warmup=5):
    """{img_s, ms_per_step, dominant kernel + roofline fraction, bs=1 p50 (HIP graph)} of one more configuration: same step as the
    default workload (forward + PostProcess on a resident synthetic batch, launch chains as LWDETR.detect chooses them)."""
    from lwdetr_amd.models import lwdetr as _lw
    T = DTYPES[dtype]
    cfg = lwdetr_amd.get_args(size)
    model, _, post = lwdetr_amd.build_model(cfg)
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).to(T).eval()
    pp = post["bbox"]
    images = synth_images(batch, res, res, seed=1234).to(dev).to(T)
    sizes = torch.tensor([[480.0, 640.0]] * batch, device=dev)
    for _ in range(warmup):
        model.detect(images, sizes, pp)
    torch.cuda.synchronize(dev)
    passes = []
    for _ in range(3):                      # three timed passes of `steps` steps; the FIRST one is reported - the methodology of the
        t0 = time.perf_counter()            # default workload's timed region (one region after the warm-up) - and all are listed
        for _ in range(steps):
            _o, det = model.detect(images, sizes, pp)
        torch.cuda.synchronize(dev)
        passes.append(time.perf_counter() - t0)
    dt = passes[0]
    assert torch.isfinite(det).all()
    out = {"workload": f"LW-DETR-{size} {res}x{res} batch {batch} {dtype}", "img_s": round(batch * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "ms_per_step_passes": [round(t / steps * 1e3, 3) for t in passes],
           "methodology": "first timed pass after the warm-up, as the default workload; later passes listed", "launch_chains": type(model)._chains_for(batch, res, res)}
    gf = GFLOP_PER_IMAGE.get((size, res))
    if gf:
        out["model_mfma_frac"] = round(out["img_s"] * gf / 1e3 / PEAK_TFLOPS[dtype], 4)
    before = _lw._STREAMS
    try:                                    # per-kernel HIP events, one launch chain (see the roofline pass of the default workload)
        _lw.set_streams(1)
        model.detect(images, sizes, pp)
        torch.cuda.synchronize(dev)
        _native.prof_enable(True)
        for _ in range(2):
            model.detect(images, sizes, pp)
        torch.cuda.synchronize(dev)
        prof = _native.prof_collect()
    finally:
        _native.prof_enable(False)
        _lw.set_streams(before)
    tot = sum(v["ms"] for v in prof.values())
    name, v = max(prof.items(), key=lambda kv: kv[1]["ms"])
    avg_ms, fl, by = v["ms"] / v["count"], v["flops"] / v["count"], v["bytes"] / v["count"]
    ridge = PEAK_TFLOPS[dtype] * 1e12 / (PEAK_HBM_GBS * 1e9)
    mfma = fl > 0 and fl / max(by, 1.0) >= ridge
    ach = fl / (avg_ms * 1e-3) / 1e12 if mfma else by / (avg_ms * 1e-3) / 1e9
    out["dominant_kernel"] = {"kernel": name, "share": round(v["ms"] / tot, 3), "avg_launch_us": round(avg_ms * 1e3, 1), "bound": "mfma" if mfma else "hbm",
                              "frac": round(ach / (PEAK_TFLOPS[dtype] if mfma else PEAK_HBM_GBS), 4)}
    one = images[:1].contiguous()
    graphed = model.capture(one, postprocess=pp, target_sizes=sizes[:1])
    lat = []
    for i in range(60):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        graphed(one)
        torch.cuda.synchronize(dev)
        if i >= 10:
            lat.append((time.perf_counter() - t) * 1e3)
    lat.sort()
    out["latency_bs1_hipgraph_ms_p50"] = round(lat[len(lat) // 2], 3)
    del graphed, model
    torch.cuda.empty_cache()
    return out


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ---- GPU telemetry beside the timed region (round 6): the boxes of this pool differ by 12-15 % on the same tree and the record
# carried no clock, power or temperature to tell a slow draw from a slow tree. Read from the card's sysfs hwmon files (no subprocess, nothing on the GPU - but milliseconds per sample:
# never between the warm-up and the timed region); `rocm-smi --json` once as the fallback. All failures are swallowed: telemetry never costs the line.
_HWMON_DIR = []          # [path or None, how it was chosen] once resolved


def _hwmon_dir():
    """hwmon directory of THE card this process computes on: the drm card whose PCI address is device 0's (a multi-GPU host shows every card in
    sysfs whatever the container may use - round 6's first version read the first AMD card it found, which on those hosts is somebody else's:
    95 MHz and 243 W 'under load'); the first AMD card only when the address cannot be matched (recorded in `card`)."""
    import glob
    if _HWMON_DIR:
        return _HWMON_DIR[0]
    want = None
    try:
        import torch
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        want = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:      # noqa: BLE001
        pass
    first = None
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if open(os.path.join(card, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        hw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*")))
        if not hw:
            continue
        addr = os.path.basename(os.path.realpath(card))
        if want and addr.startswith(want):
            _HWMON_DIR[:] = [hw[0], "pci address %s of device 0" % addr]
            return hw[0]
        first = first or (hw[0], "first AMD card in sysfs (%s; device 0 is %s)" % (addr, want))
    _HWMON_DIR[:] = list(first) if first else [None, "none"]
    return _HWMON_DIR[0]


_HWMON = {"sclk_mhz": ("freq1_input", 1e-6), "mclk_mhz": ("freq2_input", 1e-6), "power_w": ("power1_average", 1e-6), "power_input_w": ("power1_input", 1e-6),
          "temp_edge_c": ("temp1_input", 1e-3), "temp_junction_c": ("temp2_input", 1e-3), "temp_mem_c": ("temp3_input", 1e-3)}


def gpu_telemetry(hw=None):
    """{sclk_mhz, mclk_mhz, power_w, temp_*_c} right now, or {} (no readable source)."""
    out = {}
    hw = hw or _hwmon_dir()
    if hw:
        for k, (f, sc) in _HWMON.items():
            try:
                out[k] = round(int(open(os.path.join(hw, f)).read().strip()) * sc, 1)
            except (OSError, ValueError):
                pass
    if out:
        out["source"] = "sysfs hwmon"
        if _HWMON_DIR:
            out["card"] = _HWMON_DIR[1]
        return out
    try:
        import subprocess
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=10)
        card = next(iter(json.loads(r.stdout).values()))
        for k, v in card.items():
            kl = k.lower()
            num = "".join(ch for ch in str(v) if ch.isdigit() or ch == ".")
            if not num:
                continue
            if "sclk" in kl and "mhz" in str(v).lower() or kl.startswith("sclk clock speed"):
                out["sclk_mhz"] = float(num)
            elif "mclk" in kl:
                out["mclk_mhz"] = float(num)
            elif "power" in kl and "(w)" in kl:
                out["power_w"] = float(num)
            elif "temperature" in kl and "edge" in kl:
                out["temp_edge_c"] = float(num)
            elif "temperature" in kl and "junction" in kl:
                out["temp_junction_c"] = float(num)
        if out:
            out["source"] = "rocm-smi --json"
    except Exception:      # noqa: BLE001
        pass
    return out


class TelemetrySampler:
    """Samples gpu_telemetry() every few ms on a thread while a NON-reported pass runs (never during the timed region: a Python thread beside the
    launching one is not free); .stop() -> {key: [min, mean, max]} of what the card did under this workload's load."""

    def __init__(self, period=0.004):
        import threading
        self.hw, self.period, self.rows, self._stop = _hwmon_dir(), period, [], threading.Event()
        self.th = threading.Thread(target=self._run, daemon=True) if self.hw else None

    def _run(self):
        while not self._stop.is_set():
            self.rows.append(gpu_telemetry(self.hw))
            self._stop.wait(self.period)

    def start(self):
        if self.th:
            self.th.start()
        return self

    def stop(self):
        if not self.th:
            return {}
        self._stop.set()
        self.th.join(timeout=1.0)
        out = {"samples": len(self.rows)}
        for k in ("sclk_mhz", "mclk_mhz", "power_w", "power_input_w", "temp_junction_c"):
            v = [r[k] for r in self.rows if k in r]
            if v:
                out[k] = [min(v), round(sum(v) / len(v), 1), max(v)]
        return out


def main():
    a = parse()
    if a.cpu_baseline_worker:
        return _cpu_baseline_worker(a)
    if a.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return self_launch(a)                       # started plainly: become the launcher of N ranks
    rank, world, local = ldist.init_from_env(rccl_log=True)      # opt-in: config.rccl reports the channel transports
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the process group has {world} rank(s); refusing to report")
    if a.launch_check:
        return launch_check(rank, world)
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a ROCm device is required"
    if world > 1:
        assert torch.distributed.get_backend() == "nccl", "multi-GPU runs go over RCCL (backend 'nccl')"
        assert torch.cuda.device_count() > local, f"rank {rank}: local rank {local} has no GPU"
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
    # under a launcher (torchrun) the process group exists even with one rank: the collective then runs on the backend too, so
    # a one-rank torchrun of this file exercises RCCL exactly as an N-rank job does (tests/test_gpu_dist.py)
    grouped = torch.distributed.is_available() and torch.distributed.is_initialized()
    backend = torch.distributed.get_backend() if grouped else None
    gathered = torch.empty(world * a.batch, cfg.num_select, 6, dtype=torch.float32, device=dev) if grouped else None

    def step():
        _out, det = model.detect(images, sizes, pp)       # forward + PostProcess -> (B, K, 6) records
        return ldist.all_gather_detections(det, gathered, always_collective=grouped)

    def barrier():
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # telemetry sample 1: BEFORE the warm-up. (Round 6 first put it between the warm-up and the timed region: reading seven hwmon files is
    # milliseconds of idle GPU right in front of the region, and the same-box A/B against the round-5 tree showed the reported first pass 0.8 %
    # (config 2) / 1.7 % (tiny) slower while every later pass was equal - profiles/r6d_*. The instrument must not touch what it measures.)
    tele_before = gpu_telemetry() if rank == 0 else {}
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
    tele_after = gpu_telemetry() if rank == 0 else {}
    dt_nog = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
        # the same steps without the collective (outside the reported region): what the all-gather costs
        barrier()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            model.detect(images, sizes, pp)
        barrier()
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt_nog = tt.item()
    assert torch.isfinite(det).all()
    log(f"timed {a.steps} steps: {dt / a.steps * 1e3:.3f} ms/step")
    ms_step = dt / a.steps * 1e3
    ips = world * a.batch * a.steps / dt
    # three more passes of the same K steps AFTER the reported region (never part of `value`): a slow draw of the box or of the one
    # timed region shows in the record
    extra_passes = []
    tele_load = {}
    for ip in range(4):
        barrier()
        sampler = TelemetrySampler().start() if (rank == 0 and ip == 3) else None       # a FOURTH pass carries the sampling thread (clocks / power under load) and is not listed:
                                                                                        # the thread costs that pass ~1 % (profiles/r6d_*)
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step()
        barrier()
        tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if sampler:
            tele_load = sampler.stop()
        if world > 1:
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        if ip < 3:
            extra_passes.append(tt.item())

    result = {
        "metric": "images/sec", "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"LW-DETR-{a.size} inference forward + PostProcess, {a.res}x{a.res}, batch {a.batch}/GPU, "
                               f"{a.dtype}, random-init weights (synthetic COCO-shaped input)",
                   "global_batch": world * a.batch, "per_gpu_batch": a.batch, "parallelism": f"dp{world}",
                   "collective": "all_gather_into_tensor of (B,K,6) f32 detections" if grouped else "none", "backend": backend,
                   "launch_chains": type(model)._chains_for(a.batch, a.res, a.res)},
    }
    all_ms = sorted([ms_step] + [t / a.steps * 1e3 for t in extra_passes])
    result["ms_per_step_passes"] = {"timed": round(ms_step, 3), "after": [round(t / a.steps * 1e3, 3) for t in extra_passes],
                                    "median_of_all": round((all_ms[1] + all_ms[2]) / 2, 3), "median_images_per_sec": round(world * a.batch / ((all_ms[1] + all_ms[2]) / 2) * 1e3, 1),
                                    "note": "value / ms_per_step come from `timed` alone; `after` = three more passes of the same K steps outside the timed region; "
                                            "median_* = over the four (a fifth pass, not listed, carries the telemetry sampling thread)"}
    try:
        pr = torch.cuda.get_device_properties(dev)
        import hashlib, socket
        result["config"]["box"] = {"device": pr.name, "cus": pr.multi_processor_count, "mem_gb": round(pr.total_memory / 2 ** 30),
                                   "clock_mhz": getattr(pr, "clock_rate", 0) // 1000, "gcn_arch": getattr(pr, "gcnArchName", ""),
                                   "host": hashlib.sha1(socket.gethostname().encode()).hexdigest()[:8],
                                   "telemetry": {"before_warmup": tele_before, "after_timed_region": tele_after, "under_load_extra_pass": tele_load}}
    except Exception as e:                      # never fail the line over a label
        result["config"]["box"] = {"error": str(e)[:80]}
    if rank == 0 and backend == "nccl":
        result["config"]["rccl"] = ldist.rccl_report()      # RCCL version + the transports of the channels rank 0 connected
    if dt_nog is not None:
        result["ms_per_step_without_all_gather"] = round(dt_nog / a.steps * 1e3, 3)
    gf = GFLOP_PER_IMAGE.get((a.size, a.res))
    if gf:
        result["model_tflops"] = round(ips * gf / 1e3, 2)
        result["model_mfma_frac"] = round(ips * gf / 1e3 / (PEAK_TFLOPS[a.dtype] * world), 4)

    if rank == 0 and not a.no_roofline:
        # dedicated pass with per-kernel HIP events on the launch stream (outside the timed region). The timed region runs a
        # batch of >= 32 images as two launch chains on two streams (LWDETR._forward_chains); a kernel's duration measured
        # while the other chain shares the chip says nothing about the kernel, so this pass runs ONE chain: per-kernel figures
        # (and the rocprofv3 summaries under profiles/) are those of the full-batch launches on their own
        from lwdetr_amd.models import lwdetr as _lw
        streams_before = _lw._STREAMS
        _lw.set_streams(1)
        step()
        torch.cuda.synchronize(dev)
        _native.prof_enable(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize(dev)
        prof = _native.prof_collect()
        _native.prof_enable(False)
        _lw.set_streams(streams_before)
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
        # HBM bytes per launch need rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs): they are not
        # measurable from inside this process, so the figure is read from the tracked summary of those passes for
        # this workload (tools/profile_round.sh -> profiles/pmc_summary_<workload>.json) and labelled as such
        traffic, tsrc = None, None
        wl = f"{a.size}_b{a.batch}_{a.res}_{a.dtype}"
        tfp = os.path.join(ROOT, "profiles", f"pmc_summary_{wl}.json")
        if os.path.exists(tfp):
            rows = {k: v for k, v in json.load(open(tfp)).items() if k == name or k.startswith(name + "_")}
            if len(rows) == 1:                        # one rocprof kernel class behind this bench kernel name
                row = next(iter(rows.values()))
                traffic = row.get("hbm_bytes_per_launch")
                roof["mfma_busy_frac"] = row.get("mfma_busy_frac")
                tsrc = (f"profiles/pmc_summary_{wl}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES "
                        f"passes of this workload, tools/profile_round.sh, committed; not measured in this run)")
        roof.update({"kernel": name, "avg_launch_us": round(avg_ms * 1e3, 2), "launches": v["count"], "measured_as": "one launch chain, full-batch launches",
                     "alg_flops_per_launch": fl, "alg_bytes_per_launch": by, "traffic": traffic,
                     "traffic_source": tsrc})
        result["roofline"] = roof
        result["kernels"] = table

    # single-image latency (after the timed region; ~1 s): reported by default at N=1, on request otherwise
    if rank == 0 and (a.latency or world == 1) and not a.no_latency:
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
        try:
            result["latency_bs1_ms"] = p50p90(lambda: pp.select(*(lambda o: (o["pred_logits"], o["pred_boxes"]))(model(one)), sizes[:1]))
        except Exception as e:                                   # never let the extra measurement cost the bench line
            result["latency_bs1_ms"] = {"error": repr(e)[:200]}
        try:
            graphed = model.capture(one, postprocess=pp, target_sizes=sizes[:1])
            result["latency_bs1_hipgraph_ms"] = p50p90(lambda: graphed(one))
        except Exception as e:                                   # report, never hide: the eager number above stands
            result["latency_bs1_hipgraph_ms"] = {"error": repr(e)[:200]}

    default_workload = (a.size, a.batch, a.res, a.dtype) == ("small", 32, 640, "fp16")
    if rank == 0 and world == 1 and default_workload and not a.no_other_configs:
        # the other BASELINE configurations (and tiny) on this GPU, 20 steps each: config 3 as stated, configs 4 / 5 as the per-GPU shard
        result["other_configs"] = {}
        for (sz, bt, rs, dt_) in OTHER_CONFIGS:
            log(f"other config: {sz} B={bt} {rs} {dt_}")
            try:
                result["other_configs"][f"{sz}_b{bt}_{rs}_{dt_}"] = run_other_config(sz, bt, rs, dt_, dev)
            except Exception as e:                                   # never let the extra measurement cost the bench line
                result["other_configs"][f"{sz}_b{bt}_{rs}_{dt_}"] = {"error": repr(e)[:200]}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("cpu baseline (child process)")
        result["cpu_baseline"] = cpu_baseline(a.size, a.res)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if grouped:
        torch.distributed.barrier()          # rank 0 may still be in its (untimed) roofline / latency passes: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
