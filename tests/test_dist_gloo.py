"""CPU, world_size 2, gloo: batch sharding + the single all-gather of detections reproduce the single-process result."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _retry_rendezvous(fn):
    """Multi-process tests rendezvous on a probed-free TCP port; if another process grabs it in between (or the box is
    briefly too busy for the rendezvous timeout) the attempt is repeated once on a fresh port."""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        try:
            return fn(*a, **k)
        except Exception as first:          # noqa: BLE001 - any failure of the first attempt gets exactly one retry
            try:
                return fn(*a, **k)
            except Exception:
                raise first
    return wrapper


def _fake_detect(images, sizes):
    """Deterministic per-image 'detector' (the real forward needs a GPU): K=5 detections derived from the pixels."""
    b = images.shape[0]
    feat = images.reshape(b, -1)[:, :30].reshape(b, 5, 6)
    scores = feat[..., 0].sigmoid()
    labels = (feat[..., 1].abs() * 10).long() % 91
    boxes = feat[..., 2:6] * sizes[:, None, [1, 0, 1, 0]]
    return scores, labels, boxes


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lwdetr_amd import dist as D
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    images = torch.randn(8, 3, 4, 4, generator=g)
    sizes = torch.tensor([[480.0, 640.0]] * 8)
    s, l, b = D.detect_sharded(_fake_detect, images, sizes)
    q.put((rank, s, l, b))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@_retry_rendezvous
def test_sharded_detection_equals_single_process(world):
    from lwdetr_amd import dist as D
    g = torch.Generator().manual_seed(0)
    images = torch.randn(8, 3, 4, 4, generator=g)
    sizes = torch.tensor([[480.0, 640.0]] * 8)
    exp = D.unpack_detections(D.pack_detections(*_fake_detect(images, sizes)))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=120) for _ in range(world)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, s, l, b in got:
        assert torch.equal(s, exp[0]) and torch.equal(l, exp[1]) and torch.equal(b, exp[2]), rank


def test_world_size_one_is_identity_and_shards_cover_batch():
    from lwdetr_amd import dist as D
    p = torch.randn(3, 5, 6)
    assert D.all_gather_detections(p) is p
    for total, world in [(256, 8), (128, 8), (10, 4), (3, 8)]:
        r = [D.shard_range(total, k, world) for k in range(world)]
        assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert D.init_from_env() == (0, 1, 0) or os.environ.get("WORLD_SIZE", "1") != "1"


def _eval_worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lwdetr_amd import dist as D
    D.init_from_env("gloo")
    images, sizes, ids = _eval_job()
    lo, hi = D.shard_range(images.shape[0], rank, world)
    s, l, b = _fake_detect(images[lo:hi], sizes[lo:hi])
    res = D.gather_for_evaluation(ids[lo:hi], s, l, b)
    q.put((rank, None if res is None else D.to_coco_results(*res)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _eval_job():
    g = torch.Generator().manual_seed(1)
    images = torch.randn(8, 3, 4, 4, generator=g)
    sizes = torch.tensor([[480.0, 640.0]] * 8)
    ids = torch.tensor([139, 285, 632, 724, 581929, 2 ** 33 + 5, 0, 2 ** 45 + 123456789], dtype=torch.int64)   # COCO-like + huge
    return images, sizes, ids


@_retry_rendezvous
def test_gathered_evaluation_results_on_rank0_equal_single_process():
    """ids ride in the one all-gather; rank 0 ends up with exactly the COCO result list a single process
    would build (reference datasets/coco_eval.py:91-113), other ranks with nothing."""
    from lwdetr_amd import dist as D
    images, sizes, ids = _eval_job()
    s, l, b = _fake_detect(images, sizes)
    exp = D.to_coco_results(*D.unpack_detections_with_ids(D.pack_detections_with_ids(ids, s, l, b)))
    assert len(exp) == 8 * 5 and exp[0]["image_id"] == 139 and exp[-1]["image_id"] == 2 ** 45 + 123456789
    box0 = b[0, 0].tolist()
    assert exp[0]["bbox"] == pytest.approx([box0[0], box0[1], box0[2] - box0[0], box0[3] - box0[1]])
    upd = D.to_evaluator_update(ids, s, l, b)
    assert set(upd) == set(ids.tolist()) and torch.equal(upd[139]["boxes"], b[0])
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got[1] is None and got[0] == exp


# ---- bench.py launch path (`python bench.py --gpus N` must start its N ranks itself and never print n_gpus != N)
def _run_bench(args, env_extra=None, timeout=180):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


@_retry_rendezvous
def test_bench_self_launches_its_ranks():
    """Started plainly (no launcher environment) with --gpus 2, bench.py re-executes itself under torch.distributed.run
    with two ranks (gloo here: no GPU) and rank 0 reports the world size it actually joined."""
    rc, line, err = _run_bench(["--gpus", "2", "--launch-check"])
    assert rc == 0, err[-2000:]
    assert line == {"launch_check": True, "n_gpus": 2, "backend": "gloo"}


@_retry_rendezvous
def test_bench_self_launches_eight_ranks():
    """The driver's scaling run goes to N = 8: the same launch path with eight ranks (gloo, CPU), every rank's shard present in
    the gathered tensor in rank order, and the IPC mode the RCCL ranks need exported to the children."""
    rc, line, err = _run_bench(["--gpus", "8", "--launch-check"], env_extra={"OMP_NUM_THREADS": "1"}, timeout=400)
    assert rc == 0, err[-2000:]
    assert line == {"launch_check": True, "n_gpus": 8, "backend": "gloo"}


def test_ipc_mode_is_exported_before_the_gpu_runtime_starts():
    """bench.py exports HSA_ENABLE_IPC_MODE_LEGACY=0 before `import torch`, lwdetr_amd.dist.ensure_dmabuf_ipc() (called by
    init_from_env for nccl groups) does it for other launchers (dmabuf IPC: the only mode this driver has; RCCL's xGMI peer-to-peer
    setup fails without it) - never over a caller's value, and NOT as a side effect of importing the library."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    code = ("import sys; sys.path.insert(0, %r); import os; import lwdetr_amd.dist as D; a = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'); "
            "ok = D.ensure_dmabuf_ipc(); print(a, os.environ['HSA_ENABLE_IPC_MODE_LEGACY'], ok)" % ROOT)
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.split() == ["None", "0", "True"]
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "1"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120).stdout.split() == ["1", "1", "True"]
    bench_head = open(os.path.join(ROOT, "bench.py")).read()
    assert bench_head.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < bench_head.index("import torch")
    from lwdetr_amd import dist as D
    rep = D.parse_rccl_log("h:1:1 [0] NCCL INFO RCCL version 2.26.6-HEAD:x\nh:1:2 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC\n"
                           "h:1:2 [0] NCCL INFO Channel 01/0 : 0[0] -> 7[7] via P2P/IPC\n")
    assert rep["version"].startswith("2.26.6") and rep["channels_p2p"] == 2 and rep["xgmi_only"]
    assert not D.parse_rccl_log("NCCL INFO Channel 00/0 : 0[0] -> 1[1] via SHM/direct/direct")["xgmi_only"]


def test_bench_refuses_world_size_mismatch():
    """Under a launcher that started 2 ranks, `--gpus 4` must not produce a line (it used to report n_gpus of a
    different job size silently)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4",
                        "--launch-check"], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode != 0
    assert "refusing to report" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_parse_rccl_log_on_an_eight_rank_init_log_sample():
    """parse_rccl_log / the per-rank view of rccl_report on an 8-rank single-node init log in RCCL's INFO format
    (tests/golden/rccl_init_8ranks_format_sample.log - a format sample assembled from the library's format strings, see its header;
    the real two-rank log of the GPU box goes through the same parser in tests/test_gpu_dist.py), plus the SHM / NET / mixed cases."""
    from lwdetr_amd import dist as D
    text = open(os.path.join(ROOT, "tests", "golden", "rccl_init_8ranks_format_sample.log")).read()
    rep = D.parse_rccl_log(text)
    assert rep["version"] == "2.26.6-HEAD:1a2b3c4" and rep["channels_p2p"] == 8 * 2 * 2 and rep["channels_shm"] == 0 and rep["channels_net"] == 0
    assert rep["xgmi_only"]
    rank3 = "\n".join(ln for ln in text.splitlines() if " [3] " in ln)             # what rank 3's own NCCL_DEBUG_FILE holds
    r3 = D.parse_rccl_log(rank3)
    assert r3["channels_p2p"] == 4 and r3["xgmi_only"] and r3["version"].startswith("2.26.6")
    # "NET/Socket : Using ..." and "NET/Plugin" lines of the bootstrap are not channels
    assert D.parse_rccl_log("x NCCL INFO NET/Socket : Using [0]lo:127.0.0.1<0>\nx NCCL INFO NET/Plugin: none")["channels_net"] == 0
    mixed = rank3 + "\nh:1:2 [3] NCCL INFO Channel 02/0 : 3[3] -> 4[4] [send] via NET/Socket/0\n"
    assert D.parse_rccl_log(mixed)["channels_net"] == 1 and not D.parse_rccl_log(mixed)["xgmi_only"]
    shm = rank3.replace("via P2P/IPC/read", "via SHM/direct/direct")
    assert D.parse_rccl_log(shm)["channels_shm"] == 4 and not D.parse_rccl_log(shm)["xgmi_only"]
    assert D.parse_rccl_log("")["xgmi_only"] is False and D.parse_rccl_log("")["version"] is None


def test_rccl_init_log_is_opt_in():
    """init_from_env leaves NCCL_DEBUG alone and writes nothing unless rccl_log=True (advisor r4); rccl_report says so."""
    import subprocess
    code = ("import sys, os; sys.path.insert(0, %r); os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533'); "
            "os.environ.pop('NCCL_DEBUG', None); import lwdetr_amd.dist as D; D.init_from_env('gloo'); "
            "print('NCCL_DEBUG' in os.environ, D._RCCL_LOG, D.rccl_report()['log'])" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout
    assert out.strip().splitlines()[-1].startswith("False None RCCL init log not requested"), out
