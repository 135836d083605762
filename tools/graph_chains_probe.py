"""Throughput of a 32-image batch: eager one / two launch chains vs HIP graphs (one graph of the batch; two half-batch graphs
replayed on two streams). Tuning probe: does taking the launches out of the host path change the step time?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import lwdetr_amd
from lwdetr_amd.models import lwdetr as L
from lwdetr_amd.synth import synth_images, synth_state_dict


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "small"
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dt = torch.bfloat16 if size == "medium" else torch.float16
    m, _, post = lwdetr_amd.build_model(lwdetr_amd.get_args(size))
    m.load_state_dict(synth_state_dict(m.state_dict(), seed=0)); m = m.cuda().to(dt).eval()
    pp = post["bbox"]
    x = synth_images(b, 640, 640, seed=1).cuda().to(dt)
    sizes = torch.tensor([[640.0, 640.0]] * b, device="cuda")
    L.set_streams(1)
    t1 = timeit(lambda: m.detect(x, sizes, pp))
    L.set_streams(0)
    t2 = timeit(lambda: m.detect(x, sizes, pp))
    g = m.capture(x, postprocess=pp, target_sizes=sizes)
    tg1 = timeit(lambda: g(x))
    h = b // 2
    ga = m.capture(x[:h].contiguous(), postprocess=pp, target_sizes=sizes[:h])
    gb = m.capture(x[h:].contiguous(), postprocess=pp, target_sizes=sizes[h:])
    side = torch.cuda.Stream()

    def two_graphs():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gb(x[h:])
        ga(x[:h])
        cur.wait_stream(side)

    tg2 = timeit(two_graphs)
    print(f"{size} B={b}: eager 1 chain {t1:.3f} ms ({b / t1 * 1e3:.0f} img/s) | eager 2 chains {t2:.3f} ({b / t2 * 1e3:.0f}) | one graph {tg1:.3f} ({b / tg1 * 1e3:.0f}) | "
          f"two half-batch graphs on two streams {tg2:.3f} ({b / tg2 * 1e3:.0f})", flush=True)


main()
