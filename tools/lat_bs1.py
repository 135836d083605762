"""Single-image latency (forward + PostProcess as ONE HIP graph, host sync per image): p50 / p90 over 200 replays.

    python tools/lat_bs1.py [--size small] [--res 640] [--dtype fp16] [--eager]
Environment knobs of the kernels (LWDETR_*) apply as usual: the tool exists to A/B them on the latency path."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="small")
    ap.add_argument("--res", type=int, default=640)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--n", type=int, default=200)
    a = ap.parse_args()
    import torch
    import lwdetr_amd
    from lwdetr_amd.synth import synth_images, synth_state_dict
    T = {"fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    dev = torch.device("cuda:0")
    model, _, post = lwdetr_amd.build_model(lwdetr_amd.get_args(a.size))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).to(T).eval()
    pp = post["bbox"]
    one = synth_images(1, a.res, a.res, seed=1234).to(dev).to(T)
    sizes = torch.tensor([[480.0, 640.0]], device=dev)
    run = (lambda: model.detect(one, sizes, pp)) if a.eager else None
    if run is None:
        graphed = model.capture(one, postprocess=pp, target_sizes=sizes)
        run = lambda: graphed(one)
    lat = []
    for i in range(a.n + 20):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        run()
        torch.cuda.synchronize(dev)
        if i >= 20:
            lat.append((time.perf_counter() - t) * 1e3)
    lat.sort()
    plan = None
    try:
        plan = next(iter(model._plans.values())) if len(model._plans) else None
    except Exception:
        pass
    print(f"{a.size} {a.res} {a.dtype} {'eager' if a.eager else 'graph'}: p50 {lat[len(lat) // 2]:.4f} ms  p10 {lat[len(lat) // 10]:.4f}  p90 {lat[len(lat) * 9 // 10]:.4f}", flush=True)


if __name__ == "__main__":
    main()
