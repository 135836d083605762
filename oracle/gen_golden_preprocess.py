"""Writes tests/golden/preprocess.npz: seeded uint8 images and what Pillow (the library behind the reference's
SquareResize, see oracle/preprocess_ref.py) makes of them. Run in the build container:  python -m oracle.gen_golden_preprocess"""
import os

import numpy as np
import PIL
from PIL import Image

CASES = [(48, 64, 64), (37, 53, 64), (64, 64, 96), (100, 75, 64), (30, 20, 64), (97, 131, 128), (64, 40, 64), (9, 200, 32)]


def main():
    rng = np.random.default_rng(20240925)
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, (h, w, s) in enumerate(CASES):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 2:                      # smooth content as well as noise
            yy, xx = np.mgrid[0:h, 0:w]
            img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
        out[f"in_{i}"] = img
        out[f"size_{i}"] = np.array(s)
        out[f"out_{i}"] = np.asarray(Image.fromarray(img).resize((s, s), Image.BILINEAR))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "preprocess.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
