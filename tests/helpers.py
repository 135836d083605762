"""Shared test utilities: golden loading, deterministic weights, key->shape tables."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# (case name -> size, image dims) - must match oracle/gen_golden.py:CASES
CASES = {
    "tiny_640": ("tiny", [(640, 640)]),
    "small_640": ("small", [(640, 640), (640, 640)]),
    "medium_640": ("medium", [(640, 640)]),
    "large_640": ("large", [(640, 640)]),
    "xlarge_640": ("xlarge", [(640, 640)]),
    "xlarge_960": ("xlarge", [(960, 960)]),
    "tiny_192x256": ("tiny", [(192, 256), (192, 256)]),
    "small_padded": ("small", [(448, 512), (320, 384)]),
    "large_padded": ("large", [(384, 320), (256, 320)]),
}
MAX_SAMPLES = 4096


def sample_idx(n):
    if n <= MAX_SAMPLES:
        return np.arange(n)
    stride = -(-n // MAX_SAMPLES)
    stride += 1 - (stride % 2)
    return np.arange(0, n, stride)[:MAX_SAMPLES]


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)


def golden_state_dict(g, seed=0):
    """Rebuild the synthetic state dict the golden was generated with, from the key/shape table it stores."""
    from lwdetr_amd.synth import synth_param
    sd = {}
    for k, s in zip(g["sd_keys"], g["sd_shapes"]):
        shape = tuple(int(x) for x in str(s).split(",") if x != "")
        sd[str(k)] = synth_param(str(k), shape, seed)
    return sd


def case_batch(name):
    """(size, padded images (B,3,H,W), mask (B,H,W) bool) for a golden case."""
    from lwdetr_amd.synth import synth_images
    size, dims = CASES[name]
    hmax, wmax = max(d[0] for d in dims), max(d[1] for d in dims)
    full = synth_images(len(dims), hmax, wmax, seed=1234)
    mask = torch.ones(len(dims), hmax, wmax, dtype=torch.bool)
    for i, (h, w) in enumerate(dims):
        full[i, :, h:, :] = 0
        full[i, :, :, w:] = 0
        mask[i, :h, :w] = False
    return size, full, mask
