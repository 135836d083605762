"""Input side of the path on the GPU: the reference's inference transform

    Compose([SquareResize([S]), ToTensor(), Normalize(mean, std)])        (deploy/benchmark.py:273-281,
                                                                            datasets/coco.py:149-153, transforms.py:223-231, :437-443)

as one call on device-resident uint8 images. ``SquareResize`` is ``PIL.Image.resize((S, S), BILINEAR)``; the kernels in
``csrc/preproc.hip`` reproduce Pillow's fixed-point separable resampler bit for bit, and ToTensor + Normalize through a
256-entry table per channel built with the reference's own f32 operations. This module only prepares the coefficient tables
(Pillow's ``precompute_coeffs`` / ``normalize_coeffs_8bpc``, vectorised) and the per-image descriptors; there is no CPU
implementation of the transform here."""
import ctypes as C
import math

import numpy as np
import torch

from . import _native as _nat

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class ResizeImage(C.Structure):       # mirrors lwdetr_resize_image (include/lwdetr_hip.h)
    _fields_ = [("src", C.c_void_p), ("height", C.c_int), ("width", C.c_int), ("row_stride", C.c_long),
                ("xbounds_off", C.c_int), ("xcoef_off", C.c_int), ("xksize", C.c_int),
                ("ybounds_off", C.c_int), ("ycoef_off", C.c_int), ("yksize", C.c_int), ("tmp_off", C.c_long)]


def resample_tables(in_size: int, out_size: int):
    """Pillow ``precompute_coeffs`` (triangle filter over the whole axis) + ``normalize_coeffs_8bpc``, all output
    positions at once: bounds (out, 2) int32 = (first tap, tap count), coef (out, ksize) int32 (22-bit fixed point)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)            # C (int) cast truncates; operands >= 0 ...
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)                      # ... or clamped to 0 anyway
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    t = np.arange(ksize, dtype=np.float64)[None, :]
    w = np.abs((t + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(w < 1.0, 1.0 - w, 0.0)
    w = np.where(t < xmax[:, None], w, 0.0)
    ww = np.zeros((out_size, 1))
    for j in range(ksize):                  # tap by tap, as the C loop does (np.sum would add pairwise from 8 taps up)
        ww[:, 0] += w[:, j]
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    fixed = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS)).astype(np.int64).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, fixed


class SquareResizeNormalize:
    """``SquareResizeNormalize(size)(images) -> (batch (B,3,S,S) in ``dtype``, sizes (B,2) f32 = original (h, w))``.

    ``images``: a list of uint8 tensors (H, W, 3) RGB (any sizes, device or host - host tensors are uploaded first), or one
    uint8 tensor (B, H, W, 3). ``sizes`` is what the reference hands ``PostProcess`` as ``orig_target_sizes``."""

    def __init__(self, size=640, mean=IMAGENET_MEAN, std=IMAGENET_STD, dtype=torch.float16, device="cuda"):
        self.size, self.dtype, self.device = int(size), dtype, torch.device(device)
        m = torch.tensor(mean, dtype=torch.float32).view(3, 1)
        s = torch.tensor(std, dtype=torch.float32).view(3, 1)
        v = torch.arange(256, dtype=torch.float32).view(1, 256)
        self.lut = v.div(255).sub(m).div(s).contiguous().to(self.device)       # ToTensor, then F.normalize: same f32 ops
        # Resample tables are per AXIS LENGTH (the x table depends on the width only, the y table on the height only):
        # in_size -> (bounds offset, coef offset, ksize) into one device buffer (int32 elements). A COCO-style evaluation sees
        # a few hundred distinct lengths. The buffer is append-only until it is full; then it is STARTED OVER (not LRU: every
        # entry is dropped) - and grown first when one batch alone needs more than it holds, so any batch fits.
        self._axis = {}
        self._table_dev = torch.empty(1 << 20, dtype=torch.int32, device=self.device)
        self._table_len = 0
        self._retired = []                   # replaced buffers: earlier (asynchronous) launches may still read them

    def _offsets_batch(self, shapes):
        """(h, w) per image -> (xbounds, xcoef, xksize, ybounds, ycoef, yksize) offsets, all valid at the same time. The tables
        of the batch's new axis lengths are built first and uploaded with ONE copy."""
        lengths = sorted({n for hw in shapes for n in hw})
        built = {}
        for n in lengths:
            if n not in self._axis:
                bounds, coef = resample_tables(n, self.size)
                built[n] = (np.ascontiguousarray(bounds).reshape(-1), np.ascontiguousarray(coef).reshape(-1), coef.shape[1])
        need = sum(bo.size + co.size for bo, co, _ in built.values())
        if self._table_len + need > self._table_dev.numel():
            # full: start over with the tables of THIS batch (earlier launches are ordered before the upload on the stream)
            for n in lengths:
                if n not in built:
                    bounds, coef = resample_tables(n, self.size)
                    built[n] = (np.ascontiguousarray(bounds).reshape(-1), np.ascontiguousarray(coef).reshape(-1), coef.shape[1])
            need = sum(bo.size + co.size for bo, co, _ in built.values())
            if need > self._table_dev.numel():
                self._retired = [self._table_dev]
                self._table_dev = torch.empty(2 * need, dtype=torch.int32, device=self.device)
            self._axis.clear()
            self._table_len = 0
        if built:
            off, chunks = self._table_len, []
            for n, (bo, co, ks) in built.items():
                self._axis[n] = (off, off + bo.size, ks)
                chunks += [bo, co]
                off += bo.size + co.size
            flat = torch.from_numpy(np.concatenate(chunks).astype(np.int32, copy=False))
            self._table_dev[self._table_len:off].copy_(flat)
            self._table_len = off
        return [self._axis[w] + self._axis[h] for h, w in shapes]

    @torch.no_grad()
    def __call__(self, images):
        if isinstance(images, torch.Tensor) and images.dim() == 4:
            images = list(images)
        imgs = []
        for im in images:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
                raise ValueError("images must be uint8 tensors of shape (H, W, 3)")
            im = im.to(self.device)
            if im.stride(2) != 1 or im.stride(1) != 3:
                im = im.contiguous()
            imgs.append(im)
        b, s = len(imgs), self.size
        descs = (ResizeImage * max(b, 1))()
        tmp_off = 0
        offs = self._offsets_batch([(int(im.shape[0]), int(im.shape[1])) for im in imgs])
        for d, im, o in zip(descs, imgs, offs):
            h, w = int(im.shape[0]), int(im.shape[1])
            d.src, d.height, d.width, d.row_stride = im.data_ptr(), h, w, im.stride(0)
            d.xbounds_off, d.xcoef_off, d.xksize, d.ybounds_off, d.ycoef_off, d.yksize = o
            d.tmp_off = tmp_off
            tmp_off += h * s * 3
        out = torch.empty(b, 3, s, s, dtype=self.dtype, device=self.device)
        sizes = torch.tensor([[im.shape[0], im.shape[1]] for im in imgs], dtype=torch.float32, device=self.device).view(b, 2)
        if b == 0:
            return out, sizes
        tmp = torch.empty(tmp_off, dtype=torch.uint8, device=self.device)
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)      # descriptor array -> device
        with torch.cuda.device(self.device):
            rc = _nat.lib().lwdetr_resize_normalize(raw.data_ptr(), b, max(int(im.shape[0]) for im in imgs),
                                                    self._table_dev.data_ptr(), tmp.data_ptr(), self.lut.data_ptr(),
                                                    out.data_ptr(), s, _nat.dtype_code(self.dtype), _nat.stream_ptr(self.device))
        _nat.check(rc, "lwdetr_resize_normalize")
        self._keep = (imgs, tmp, raw)           # alive until the next call (the launch is asynchronous)
        return out, sizes


def infer_transforms(size=640, dtype=torch.float16, device="cuda"):
    """The reference's ``infer_transforms()`` (deploy/benchmark.py:273-281) for device-resident uint8 images."""
    return SquareResizeNormalize(size, dtype=dtype, device=device)
