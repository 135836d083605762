"""Multi-GPU batched inference: one process per GPU, images sharded contiguously, ONE fixed-shape all-gather of the
detections (RCCL over xGMI on a node; ``gloo`` in the CPU tests).

Replaces the reference's distributed evaluation plumbing - ``DistributedSampler`` sharding (``main.py:222-224``) and the
pickle -> ByteTensor -> padded ``all_gather`` -> unpickle of per-image results (``util/misc.py:99-139``,
``datasets/coco_eval.py:181-200``) - with a single ``all_gather_into_tensor`` of a ``(B/W, K, 6)`` f32 tensor
(score, label, x0, y0, x1, y1; labels < 2^24 are exact in f32). No other collective is on the data path.
"""
import glob
import os
import re
import shutil
import sys
import tempfile

import torch
import torch.distributed as dist

_RCCL_LOG = None        # NCCL_DEBUG_FILE pattern of this job when init_from_env(rccl_log=True) switched the RCCL init log on
_RCCL_LOG_DIR = None    # its private directory (removed by rccl_report)


def ensure_dmabuf_ipc():
    """The ROCm host driver of the MI355X pool supports dmabuf IPC only: RCCL's intra-node peer-to-peer setup over xGMI
    (hipIpcGetMemHandle / hipIpcOpenMemHandle between the ranks of a node) fails with "invalid argument" in the legacy IPC mode.
    The HSA runtime reads HSA_ENABLE_IPC_MODE_LEGACY when it starts (the first device call of the process), so the default must be
    in place BEFORE that: launchers (bench.py) call this - or export the variable - before their first device call; init_from_env()
    calls it for ``nccl`` groups and warns when the runtime was already up. Never overrides a value the caller chose. Returns True
    when the variable is in place in time (already exported, or set now with the GPU runtime still down)."""
    if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:
        return True
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return not (torch.cuda.is_available() and torch.cuda.is_initialized())


def init_from_env(backend=None, single_node=None, force=None, rccl_log=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank). Without a launcher environment (no RANK) a single process is a no-op returning
    (0, 1, 0); under a launcher the process group is created even for WORLD_SIZE=1 (``force`` overrides either way), so a
    one-rank ``torchrun`` exercises the same backend initialisation (RCCL on a GPU box) as an N-rank job.
    ``single_node`` (default: inferred from LOCAL_WORLD_SIZE == WORLD_SIZE) gates the loopback-only RCCL bootstrap.
    ``rccl_log`` (opt-in; bench.py): route RCCL's init log (NCCL_DEBUG=INFO) of this process into a private temporary directory so
    that rccl_report() can say which transports the channels use; the directory is removed by rccl_report(). Nothing is written
    and NCCL_DEBUG is left alone otherwise, or when the caller already set NCCL_DEBUG."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if force is None:
        force = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world <= 1 and not force:
        return 0, 1, 0
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"      # "nccl" is RCCL on ROCm
    if backend == "nccl":
        if single_node is None:     # torchrun exports LOCAL_WORLD_SIZE: all ranks on this node <=> it equals WORLD_SIZE
            single_node = int(os.environ.get("LOCAL_WORLD_SIZE", "0")) == world
        if single_node:
            # one node, xGMI only: keep RCCL's bootstrap on loopback and off InfiniBand. Only applied when the job is
            # known to be single-node (on a multi-node job these would force a loopback bootstrap and hang), and never
            # over a value the caller has set.
            for k, v in (("NCCL_IB_DISABLE", "1"), ("NCCL_SOCKET_IFNAME", "lo")):
                if k not in os.environ:
                    os.environ[k] = v
                    print(f"[lwdetr_amd.dist] single-node job: {k}={v}", file=sys.stderr, flush=True)
        if not ensure_dmabuf_ipc():
            print("[lwdetr_amd.dist] warning: the GPU runtime was initialised before HSA_ENABLE_IPC_MODE_LEGACY was exported; "
                  "call lwdetr_amd.dist.ensure_dmabuf_ipc() (or export HSA_ENABLE_IPC_MODE_LEGACY=0) before the first device call "
                  "for multi-rank RCCL on this driver (dmabuf IPC only)", file=sys.stderr, flush=True)
        if rccl_log and "NCCL_DEBUG" not in os.environ:
            # RCCL's init log (version, topology, the transport of every channel) into a per-process file: rccl_report() reads
            # this process's copy after the run so that the bench line can show "P2P over xGMI only". INFO logs at init / connect only.
            global _RCCL_LOG, _RCCL_LOG_DIR
            _RCCL_LOG_DIR = tempfile.mkdtemp(prefix="lwdetr_rccl_")
            _RCCL_LOG = os.path.join(_RCCL_LOG_DIR, "rccl")
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,P2P,SHM,NET")
            os.environ["NCCL_DEBUG_FILE"] = _RCCL_LOG + ".%p.log"
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def parse_rccl_log(text):
    """RCCL / NCCL INFO log -> {"version", "channels_p2p", "channels_shm", "channels_net", "xgmi_only"}: the `... via P2P/IPC`
    (xGMI / PCIe peer access), `via SHM` (host memory bounce) and `via NET` (sockets / IB) channel lines of the connect phase."""
    ver = re.search(r"(?:RCCL|NCCL) version ([0-9][^\s]*)", text)
    chan = [ln for ln in text.splitlines() if re.search(r"\bChannel \d+(?:/\d+)? : ", ln)]     # connect-phase lines only
    p2p = sum(1 for ln in chan if "via P2P/" in ln)
    shm = sum(1 for ln in chan if "via SHM" in ln)
    net = sum(1 for ln in chan if "via NET/" in ln)
    return {"version": ver.group(1) if ver else None, "channels_p2p": p2p, "channels_shm": shm, "channels_net": net,
            "xgmi_only": bool(p2p) and not shm and not net}


def rccl_report():
    """What this process's RCCL saw (bench.py prints it for N > 1): library version as torch reports it, and - when
    init_from_env(rccl_log=True) switched the init log on - the transports of the channels this rank connected. Removes the
    temporary log directory (call it once, after the collectives have run)."""
    global _RCCL_LOG, _RCCL_LOG_DIR
    rep = {"torch_nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if torch.cuda.is_available() else None,
           "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_SOCKET_IFNAME", "NCCL_IB_DISABLE", "NCCL_P2P_DISABLE")}}
    if not _RCCL_LOG:
        rep["log"] = (f"NCCL_DEBUG={os.environ.get('NCCL_DEBUG')} was set by the caller: RCCL's init log is where the caller sent it"
                      if os.environ.get("NCCL_DEBUG") else "RCCL init log not requested (init_from_env(rccl_log=True))")
        return rep
    text = ""
    for fp in glob.glob(_RCCL_LOG + f".{os.getpid()}.log"):
        try:
            text += open(fp, errors="replace").read()
        except OSError:
            pass
    rep.update(parse_rccl_log(text))
    rep["log"] = "NCCL_DEBUG=INFO init log of rank 0" if text else "no RCCL log found"
    shutil.rmtree(_RCCL_LOG_DIR, ignore_errors=True)
    _RCCL_LOG = _RCCL_LOG_DIR = None
    return rep


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


def all_gather_detections(packed, out=None, always_collective=False):
    """All ranks contribute (b, K, 6) with the SAME b; returns (world*b, K, 6) in rank order. A group of one rank returns its
    input unless ``always_collective`` (tests: the collective itself runs on the backend, e.g. RCCL with one GPU)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not always_collective):
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


# ------------------------------------------------------------------------------------------------------------------
# The step after the path: hand the gathered detections to the COCO evaluator on rank 0 only.
# The reference updates a CocoEvaluator on every rank and merges the per-rank pycocotools state through pickled
# all_gathers (datasets/coco_eval.py:56-69, :181-200, engine.py:142-157); with the detections already gathered as one
# fixed-shape tensor, rank 0 can run the stock evaluator on the full result set and no evaluator state crosses ranks.
_ID_BITS = 21                                    # image ids travel as three 21-bit limbs: exact in f32, ids < 2^63


def pack_detections_with_ids(image_ids, scores, labels, boxes):
    """(B,) int64 ids + detections -> (B, K+1, 6) f32: row 0 carries the id (three 21-bit limbs), rows 1.. the detections.
    The ids ride in the same all-gather as the detections - still ONE collective on the data path."""
    ids = image_ids.to(torch.int64).to(scores.device)
    if bool((ids < 0).any()):
        raise ValueError("image ids must be non-negative")
    mask = (1 << _ID_BITS) - 1
    head = torch.zeros(ids.shape[0], 1, 6, dtype=torch.float32, device=scores.device)
    head[:, 0, 0] = (ids & mask).float()
    head[:, 0, 1] = ((ids >> _ID_BITS) & mask).float()
    head[:, 0, 2] = (ids >> (2 * _ID_BITS)).float()
    return torch.cat([head, pack_detections(scores, labels, boxes)], 1).contiguous()


def unpack_detections_with_ids(packed):
    head = packed[:, 0].to(torch.int64)
    ids = head[:, 0] | (head[:, 1] << _ID_BITS) | (head[:, 2] << (2 * _ID_BITS))
    return (ids,) + tuple(unpack_detections(packed[:, 1:]))


def to_coco_results(image_ids, scores, labels, boxes):
    """-> the list of dicts ``CocoEvaluator.prepare_for_coco_detection`` builds (datasets/coco_eval.py:91-113):
    xyxy -> xywh (``convert_to_xywh``, :176-178), python scalars. Accepts the gathered tensors of the whole job."""
    xywh = torch.stack([boxes[..., 0], boxes[..., 1], boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]], -1)
    ids, sc, lb, bx = image_ids.tolist(), scores.tolist(), labels.tolist(), xywh.tolist()
    return [{"image_id": ids[i], "category_id": lb[i][k], "bbox": bx[i][k], "score": sc[i][k]}
            for i in range(len(ids)) for k in range(len(sc[i]))]


def to_evaluator_update(image_ids, scores, labels, boxes):
    """-> ``{image_id: {"scores", "labels", "boxes"}}``, the ``res`` dict ``engine.evaluate`` passes to
    ``coco_evaluator.update`` (engine.py:153-155)."""
    return {int(i): {"scores": s, "labels": l, "boxes": b} for i, s, l, b in zip(image_ids.tolist(), scores, labels, boxes)}


def gather_for_evaluation(image_ids, scores, labels, boxes, always_collective=False):
    """Every rank contributes its shard (same b on every rank); returns the whole job's (ids, scores, labels, boxes) on
    rank 0 and None elsewhere. Feed rank 0's result to ``to_evaluator_update`` / ``to_coco_results``."""
    full = all_gather_detections(pack_detections_with_ids(image_ids, scores, labels, boxes), always_collective=always_collective)
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    return unpack_detections_with_ids(full) if rank == 0 else None
