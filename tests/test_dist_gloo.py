"""CPU, world_size 2, gloo: batch sharding + the single all-gather of detections reproduce the single-process result."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
