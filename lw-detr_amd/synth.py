"""Deterministic synthetic weights and inputs (no checkpoints / datasets are reachable offline).

Everything is derived from a counter-based integer hash (splitmix64) evaluated with numpy uint64
arithmetic, so the same name/shape/seed yields bit-identical tensors on any machine - the golden
fixtures under ``tests/golden`` were produced by loading exactly these tensors into the reference
model, and the tests/bench regenerate them instead of shipping a checkpoint.

The value ranges deliberately break the reference's degenerate initialisation (SURVEY.md section 8c:
``refpoint_embed`` = 0 ``models/lwdetr.py:69``, last bbox layer = 0 ``:90-91``, ``attention_weights`` = 0
and ``sampling_offsets.weight`` = 0 ``models/ops/modules/ms_deform_attn.py:80-90``, BN stats (0, 1)), so that
every arithmetic branch of the forward path is exercised.
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(key: int, n: int) -> np.ndarray:
    """n float64 values in [0, 1), a pure function of (key, index)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([key & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
        x = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + base
    z = _splitmix64(x)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _key(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) << 20) ^ (seed * 0x9E3779B1)


def uniform(name: str, shape, lo: float, hi: float, seed: int = 0) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(_key(name, seed), n)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32).reshape(tuple(shape)))


def synth_param(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Synthetic value of one state-dict entry, chosen by the entry's role (see module docstring)."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        return uniform(name, shape, 0.5, 1.5, seed)
    if leaf == "running_mean":
        return uniform(name, shape, -0.2, 0.2, seed)
    if leaf in ("gamma_1", "gamma_2"):
        return uniform(name, shape, 0.05, 0.5, seed)
    if leaf == "pos_embed":
        return uniform(name, shape, -0.5, 0.5, seed)
    if leaf in ("q_bias", "v_bias"):
        return uniform(name, shape, -0.2, 0.2, seed)
    if name.endswith("refpoint_embed.weight"):
        return uniform(name, shape, -0.5, 0.5, seed)
    if name.endswith("query_feat.weight"):
        return uniform(name, shape, -1.0, 1.0, seed)
    if "sampling_offsets.bias" in name:
        return uniform(name, shape, -2.0, 2.0, seed)
    if "sampling_offsets.weight" in name:
        a = 0.5 * (3.0 / shape[1]) ** 0.5
        return uniform(name, shape, -a, a, seed)
    if "class_embed" in name and leaf == "bias":
        return uniform(name, shape, -5.1, -4.1, seed)
    is_norm = (".norm" in name or ".bn." in name or "enc_output_norm" in name
               or (len(shape) == 1 and leaf == "weight"))
    if leaf == "weight" and is_norm and len(shape) == 1:
        return uniform(name, shape, 0.7, 1.3, seed)
    if leaf == "bias":
        return uniform(name, shape, -0.1, 0.1, seed)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        if "stages_sampling" in name and len(shape) == 4 and shape[2] == 2:   # ConvTranspose2d (Cin,Cout,2,2)
            fan_in = shape[0]
        a = (3.0 / fan_in) ** 0.5
        return uniform(name, shape, -a, a, seed)
    return uniform(name, shape, -0.1, 0.1, seed)


def synth_state_dict(template: dict, seed: int = 0) -> dict:
    """A full state dict for any module: ``template`` maps key -> tensor (only key/shape are used)."""
    return {k: synth_param(k, v.shape, seed) for k, v in template.items()}


def synth_images(batch: int, height: int, width: int, seed: int = 1234) -> torch.Tensor:
    """'COCO-shaped' normalised RGB: roughly unit-scale values, (B, 3, H, W) float32."""
    u = uniform01(_key("images", seed), batch * 3 * height * width + 1)
    # sum of two uniforms, centred: triangular distribution with std ~= 1
    t = (u[:-1] + u[1:] - 1.0) * 2.45
    return torch.from_numpy(t.astype(np.float32).reshape(batch, 3, height, width))
