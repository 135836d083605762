"""GPU: the RCCL path of the multi-GPU plan on the one GPU of the test box ("backend nccl, set_device ordering,
the loopback bootstrap defaults and all_gather_into_tensor on device tensors have never executed"). One rank is started with the
driver's launch contract (python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(args, timeout=600, nproc=1, env_extra=None):
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)      # bench.py / lwdetr_amd.dist set it themselves (dmabuf IPC)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_rccl_all_gather_of_detections_one_rank():
    r = _torchrun([os.path.join(ROOT, "tests", "rccl_one_rank_driver.py")])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL-OK")]
    assert line and "backend=nccl" in line[0], r.stdout[-2000:]
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl_one_rank.log"), "w") as f:
        f.write(line[0] + "\n" + "\n".join(ln for ln in r.stderr.splitlines() if "lwdetr_amd.dist" in ln) + "\n")


def test_bench_under_torchrun_one_rank_initialises_rccl():
    """bench.py launched the way the driver launches it for N > 1, with N = 1: the process group is created on the nccl backend and
    the line it prints says so."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--size", "tiny", "--batch", "4", "--res", "256", "--steps", "3", "--warmup", "1",
                   "--no-cpu-baseline", "--no-latency", "--no-roofline"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"].get("backend") == "nccl", d["config"]
    rccl = d["config"].get("rccl")
    assert rccl and rccl["torch_nccl_version"] and rccl["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0", rccl
    # RCCL's own init log is parsed unless the box's environment already routes NCCL_DEBUG somewhere (then the report says so)
    assert rccl.get("version") or "set by the caller" in rccl.get("log", ""), rccl
    with open(os.path.join(ROOT, "gpurun_out", "bench_torchrun_one_rank.json"), "w") as f:
        f.write(json.dumps(d) + "\n")


def test_two_ranks_on_the_one_gpu_is_recorded():
    """everything about an N > 1 `nccl` group that can run on a one-GPU box. Two ranks, both on GPU 0, through
    init_from_env + all_gather_into_tensor: RCCL either serves them (then the gathered tensor is checked and the channel transports
    of a real two-rank init log go through parse_rccl_log) or refuses a communicator with two ranks on one device - the exact
    refusal is recorded (gpurun_out/rccl_two_ranks_one_gpu.log -> profiles/). Anything else (a hang, another error) fails."""
    try:
        r = _torchrun([os.path.join(ROOT, "tests", "rccl_two_ranks_one_gpu_driver.py")], timeout=240, nproc=2,
                      env_extra={"CUDA_VISIBLE_DEVICES": "0", "HIP_VISIBLE_DEVICES": "0"})
    except subprocess.TimeoutExpired as e:
        pytest.fail("two ranks on one GPU: no answer in 240 s (hang): " + str(e.stderr)[-1500:])
    text = r.stdout + r.stderr
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL2-OK")]
    refused = [ln for ln in text.splitlines() if "Duplicate GPU detected" in ln or "invalid usage" in ln.lower() or "ncclInvalidUsage" in ln]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "rccl_two_ranks_one_gpu.log"), "w") as f:
        f.write(f"returncode {r.returncode}\n" + "\n".join(ok) + "\n--- refusal lines ---\n" + "\n".join(refused[:8]) + "\n--- stderr tail ---\n" + r.stderr[-3000:] + "\n")
    if r.returncode == 0:
        assert len(ok) == 2, text[-3000:]
    else:
        assert refused, text[-4000:]
