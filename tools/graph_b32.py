"""Tuning tool: the throughput step (dense batch, forward + PostProcess) replayed as ONE HIP graph against the eager two-chain step.
    python tools/graph_b32.py [--size small] [--batch 32]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="small")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--res", type=int, default=640)
    a = ap.parse_args()
    import torch
    import lwdetr_amd
    from lwdetr_amd.synth import synth_images, synth_state_dict
    dev = torch.device("cuda:0")
    model, _, post = lwdetr_amd.build_model(lwdetr_amd.get_args(a.size))
    model.load_state_dict(synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).half().eval()
    pp = post["bbox"]
    x = synth_images(a.batch, a.res, a.res, seed=1).to(dev).half()
    sizes = torch.tensor([[a.res, a.res]] * a.batch, device=dev, dtype=torch.float32)

    def timed(fn, steps=20, reps=4):
        out = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            out.append(round((time.perf_counter() - t0) / steps * 1e3, 3))
        return out

    def eager():
        o = model(x)
        return pp.select(o["pred_logits"], o["pred_boxes"], sizes)
    for _ in range(5):
        eager()
    print("eager (default chains) ms/step:", timed(eager))
    g = model.capture(x, postprocess=pp, target_sizes=sizes)
    for _ in range(5):
        g(x)
    print("one HIP graph (one chain) ms/step:", timed(lambda: g(x)))
    h = a.batch // 2
    g1 = model.capture(x[:h], postprocess=pp, target_sizes=sizes[:h])
    g2 = model.capture(x[h:], postprocess=pp, target_sizes=sizes[h:])
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def two():
        cur = torch.cuda.current_stream(dev)
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            g1(x[:h])
        with torch.cuda.stream(s2):
            g2(x[h:])
        cur.wait_stream(s1); cur.wait_stream(s2)
    for _ in range(5):
        two()
    print("two HIP graphs on two streams (half batches) ms/step:", timed(two))


if __name__ == "__main__":
    main()
