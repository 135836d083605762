"""Throughput of the input-side kernels (tuning / measurement tool): COCO-like 480x640 uint8 frames -> 640x640 fp16 NCHW."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lwdetr_amd.preprocess import SquareResizeNormalize  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    b = 32
    frames = [torch.randint(0, 256, (480, 640, 3), dtype=torch.uint8, device=dev) for _ in range(b)]
    tf = SquareResizeNormalize(640, dtype=torch.float16, device=dev)
    for _ in range(3):
        tf(frames)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        tf(frames)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    # algorithmic bytes: width == S, so no horizontal pass: read the source once, write the fp16 planes
    byt = b * (480 * 640 * 3 + 3 * 640 * 640 * 2)
    print(f"preprocess: batch {b} of 480x640 -> 640x640 fp16: {ms * 1e3:.1f} us / batch = {b / ms * 1e3:.0f} img/s, "
          f"{byt / ms / 1e6:.0f} GB/s algorithmic ({byt / ms / 1e6 / 8000:.1%} of 8 TB/s) [includes host-side descriptor upload]")


if __name__ == "__main__":
    main()
