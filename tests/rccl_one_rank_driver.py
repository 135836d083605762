"""Helper of tests/test_gpu_dist.py: ONE rank under ``python -m torch.distributed.run --nproc-per-node 1``. Initialises the
``nccl`` (= RCCL) backend through lwdetr_amd.dist.init_from_env exactly as an N-rank job does, runs model -> PostProcess.select_packed
-> all_gather_detections / gather_for_evaluation on device tensors with the collective forced, and checks that the gathered
result equals the local one. Prints one line starting with RCCL-OK on success."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import lwdetr_amd
    from lwdetr_amd import dist as D
    from lwdetr_amd.synth import synth_images, synth_state_dict
    rank, world, local = D.init_from_env()
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl", (rank, world, dist.get_backend())
    dev = torch.device("cuda", local)
    model, _, post = lwdetr_amd.build_model(lwdetr_amd.get_args("tiny"))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).half().eval()
    images = synth_images(2, 192, 256, seed=5).to(dev).half()
    sizes = torch.tensor([[480.0, 640.0]] * 2, device=dev)
    with torch.no_grad():
        out = model(images)
        packed = post["bbox"].select_packed(out["pred_logits"], out["pred_boxes"], sizes)
    assert packed.is_cuda and packed.dtype == torch.float32 and packed.shape[-1] == 6
    full = D.all_gather_detections(packed, always_collective=True)          # all_gather_into_tensor on RCCL, device tensors
    torch.cuda.synchronize()
    assert full.data_ptr() != packed.data_ptr() and torch.equal(full, packed)
    s, l, b = D.unpack_detections(packed)
    ids = torch.tensor([17, 2 ** 40 + 5], dtype=torch.int64, device=dev)
    got = D.gather_for_evaluation(ids, s, l, b, always_collective=True)
    assert got is not None and torch.equal(got[0], ids) and torch.equal(got[1], s) and torch.equal(got[2], l) and torch.equal(got[3], b)
    res = D.to_evaluator_update(*got)
    assert sorted(res) == [17, 2 ** 40 + 5] and res[17]["boxes"].shape == (s.shape[1], 4)
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    libs = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libnccl" in ln]
    print(f"RCCL-OK backend={dist.get_backend()} version={ver} lib={sorted(set(libs))[:1]} gathered={tuple(full.shape)}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
