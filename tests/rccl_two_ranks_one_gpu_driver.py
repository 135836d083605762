"""Helper of tests/test_gpu_dist.py: one of TWO ranks under ``python -m torch.distributed.run --nproc-per-node 2`` that both use GPU 0
(the test box has one GPU). Goes through lwdetr_amd.dist.init_from_env on the ``nccl`` (= RCCL) backend and runs the data path's
only collective - all_gather_into_tensor of a (b, K, 6) f32 tensor - between the two processes. RCCL either serves two ranks on one
device or refuses the communicator ("Duplicate GPU detected"); the test records which. Prints ``RCCL2-OK ...`` on success."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from lwdetr_amd import dist as D
    os.environ["LOCAL_RANK"] = "0"                   # both ranks on the box's one GPU
    rank, world, local = D.init_from_env(rccl_log=True)
    assert world == 2 and dist.get_backend() == "nccl"
    dev = torch.device("cuda", 0)
    packed = torch.full((4, 300, 6), float(rank + 1), device=dev)
    full = D.all_gather_detections(packed)
    torch.cuda.synchronize()
    assert full.shape == (8, 300, 6) and bool((full[:4] == 1).all()) and bool((full[4:] == 2).all())
    rep = D.rccl_report()
    print(f"RCCL2-OK rank={rank} version={rep.get('version')} p2p={rep.get('channels_p2p')} shm={rep.get('channels_shm')} "
          f"net={rep.get('channels_net')} xgmi_only={rep.get('xgmi_only')}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
