"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's inference input transform.

Reference call sites: ``datasets/transforms.py:223-231`` (``SquareResize`` -> ``torchvision.transforms.functional.resize`` of
a PIL image = ``PIL.Image.resize((S, S), BILINEAR)``), ``datasets/transforms.py:437-443`` / ``datasets/coco.py:127-130``
(``ToTensor`` + ``Normalize(mean, std)``), ``deploy/benchmark.py:273-281`` (``infer_transforms``).

The resize arithmetic lives in a third-party dependency that is NOT under /root/reference: Pillow (pulled in through
torchvision, unpinned in ``requirements.txt``; 12.2.0 in this image), ``src/libImaging/Resample.c``:
``precompute_coeffs`` (triangle filter, support scaled by the down-scale factor, coefficients normalised in double),
``normalize_coeffs_8bpc`` (fixed point, PRECISION_BITS = 32 - 8 - 2 = 22, round half away from zero),
``ImagingResampleHorizontal_8bpc`` / ``ImagingResampleVertical_8bpc`` (accumulator starts at 1 << 21, result
``clip8(acc >> 22)``), horizontal pass first, uint8 intermediate. Pinned bit-exactly against Pillow itself
(``tests/golden/preprocess.npz``, written by ``oracle/gen_golden_preprocess.py``; ``tests/test_preprocess_oracle.py``).
Only tests/, ``__graft_entry__.smoke()`` and bench.py's cpu_baseline may import this module."""
import math

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle) filter over the whole axis.
    Returns bounds (out_size, 2) int32 = (first input index, tap count) and coefficients (out_size, ksize) int32."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - w if w < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64)
    return bounds, fixed.astype(np.int32)


def _resample_axis(img, out_size, axis):
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    bounds, k = resample_coeffs(a.shape[0], out_size)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(k[xx, :n].astype(np.int64), a[xmin:xmin + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def square_resize_u8(img, size):
    """img (H, W, 3) uint8 -> (size, size, 3) uint8, == PIL.Image.fromarray(img).resize((size, size), BILINEAR)."""
    out = np.ascontiguousarray(img)
    if out.shape[1] != size:
        out = _resample_axis(out, size, 1)      # horizontal pass first (ImagingResample)
    if out.shape[0] != size:
        out = _resample_axis(out, size, 0)
    return out


def to_tensor_normalize(u8, dtype=torch.float32, mean=MEAN, std=STD):
    """(S, S, 3) uint8 -> (3, S, S): ToTensor (/255 in f32) then Normalize ((x - mean) / std in f32), then the model dtype."""
    x = torch.from_numpy(np.ascontiguousarray(u8)).permute(2, 0, 1).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return x.sub(m).div(s).to(dtype)


def preprocess(images, size, dtype=torch.float32):
    """list of (H, W, 3) uint8 arrays -> (B, 3, size, size) tensor + (B, 2) original (h, w)."""
    out = torch.stack([to_tensor_normalize(square_resize_u8(np.asarray(im), size), dtype) for im in images])
    sizes = torch.tensor([[im.shape[0], im.shape[1]] for im in images], dtype=torch.float32)
    return out, sizes
