"""Multi-GPU batched inference: one process per GPU, images sharded contiguously, ONE fixed-shape all-gather of the
detections (RCCL over xGMI on a node; ``gloo`` in the CPU tests).

Replaces the reference's distributed evaluation plumbing - ``DistributedSampler`` sharding (``main.py:222-224``) and the
pickle -> ByteTensor -> padded ``all_gather`` -> unpickle of per-image results (``util/misc.py:99-139``,
``datasets/coco_eval.py:181-200``) - with a single ``all_gather_into_tensor`` of a ``(B/W, K, 6)`` f32 tensor
(score, label, x0, y0, x1, y1; labels < 2^24 are exact in f32). No other collective is on the data path.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank); a no-op returning (0, 1, 0) when WORLD_SIZE is absent or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" is RCCL on ROCm
    if backend == "nccl":
        # single node, xGMI only: no InfiniBand / socket transport for data
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of ``total`` images for ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_detections(scores, labels, boxes):
    """(B,K), (B,K) int64, (B,K,4) -> (B,K,6) f32 contiguous."""
    return torch.cat([scores.float().unsqueeze(-1), labels.float().unsqueeze(-1), boxes.float()], -1).contiguous()


def unpack_detections(packed):
    return packed[..., 0], packed[..., 1].long(), packed[..., 2:6]


def all_gather_detections(packed, out=None):
    """All ranks contribute (b, K, 6) with the SAME b; returns (world*b, K, 6) in rank order (identity for world 1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return packed
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed)
    return out


def detect_sharded(detect_fn, images, target_sizes):
    """Run ``detect_fn(images, target_sizes) -> (scores, labels, boxes)`` on this rank's shard of a replicated batch
    and return the full batch's detections on every rank. The batch must divide evenly (pad upstream otherwise)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    total = images.shape[0]
    if total % world:
        raise ValueError(f"batch {total} does not divide over {world} ranks")
    lo, hi = shard_range(total, rank, world)
    s, l, b = detect_fn(images[lo:hi], target_sizes[lo:hi])
    return unpack_detections(all_gather_detections(pack_detections(s, l, b)))
