"""Deterministic synthetic inputs and weights (no dependence on torch / numpy RNG streams).

Every tensor is a pure function of (seed, name, shape): a counter-based integer hash (splitmix64
finaliser) mapped to floats, so the oracle side, the HIP side and the golden-vector generator
regenerate bit-identical tensors and only OUTPUTS need to be committed as fixtures.

`fill_pose_state_dict` / `fill_flow_state_dict` write a NON-degenerate weight set: the reference's
default init gives |heatmap| ~ 1e-4 (deconv / heatmap std 0.001, pose_deconv.py:53,62), which would
make a 1e-3 tolerance vacuous and the arg-max a coin flip.
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np
import torch

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _key(seed: int, name: str, stream: int) -> np.uint64:
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    return np.uint64(((seed & 0xFFFF) << 48) ^ (h << 16) ^ (stream & 0xFFFF))


def uniform01(seed: int, name: str, shape, stream: int = 0) -> np.ndarray:
    """float64 in [0, 1), shape `shape`."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + _mix(np.array([_key(seed, name, stream)], dtype=np.uint64))
    bits = _mix(ctr)
    return ((bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)


def uniform(seed: int, name: str, shape, lo: float = 0.0, hi: float = 1.0) -> torch.Tensor:
    return torch.from_numpy((lo + (hi - lo) * uniform01(seed, name, shape)).astype(np.float32))


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> torch.Tensor:
    u1 = uniform01(seed, name, shape, stream=1)
    u2 = uniform01(seed, name, shape, stream=2)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy((mean + std * z).astype(np.float32))


# ---- inputs ---------------------------------------------------------------------------------------
def pose_crops(seed: int, B: int, H: int = 256, W: int = 192) -> torch.Tensor:
    """[B,3,H,W] ~ N(0,1): the statistics of a mean/std-normalised crop (coco.py:152)."""
    return normal(seed, "pose_crops", (B, 3, H, W))


def frame_pairs(seed: int, B: int, H: int = 384, W: int = 512, max_shift: int = 8) -> torch.Tensor:
    """[B,3,2,H,W] in [0,255]: smooth-ish random texture; frame 1 = frame 0 translated by a known
    integer shift (|s| <= max_shift, per sample) plus N(0,2) noise, so the flow is meaningful."""
    pad = max_shift
    lo = uniform01(seed, "frame_lo", (B, 3, (H + 2 * pad) // 8 + 2, (W + 2 * pad) // 8 + 2))
    base = np.kron(lo, np.ones((1, 1, 8, 8)))[:, :, :H + 2 * pad, :W + 2 * pad]
    fine = uniform01(seed, "frame_fine", (B, 3, H + 2 * pad, W + 2 * pad))
    canvas = 255.0 * (0.7 * base + 0.3 * fine)
    shifts = np.floor(uniform01(seed, "frame_shift", (B, 2)) * (2 * max_shift + 1)).astype(np.int64) - max_shift
    noise = normal(seed, "frame_noise", (B, 3, H, W), std=2.0).numpy().astype(np.float64)
    out = np.empty((B, 3, 2, H, W), dtype=np.float32)
    for b in range(B):
        sy, sx = int(shifts[b, 0]), int(shifts[b, 1])
        out[b, :, 0] = canvas[b, :, pad:pad + H, pad:pad + W]
        out[b, :, 1] = np.clip(canvas[b, :, pad + sy:pad + sy + H, pad + sx:pad + sx + W] + noise[b], 0.0, 255.0)
    return torch.from_numpy(out)


def flow_field(seed: int, B: int, H: int, W: int, magnitude: float = 6.0) -> torch.Tensor:
    """[B,2,H,W] fp32 sub-pixel flow for warp tests, including out-of-frame targets."""
    return normal(seed, "flow_field", (B, 2, H, W), std=magnitude)


def peaked_heatmaps(seed: int, N: int, K: int = 17, h: int = 64, w: int = 48, sigma: float = 2.0, noise: float = 1e-3) -> torch.Tensor:
    """[N,K,h,w] heat maps shaped like a TRAINED pose net's output: one Gaussian bump per map with the training target's sigma
    (lib/pose/utils/heatmap.py:19-60: exp(-d^2 / (2 sigma^2)), sigma = 2 map pixels, tools/pose/config.py:85), amplitude
    0.7 .. 1.0, its centre at a uniformly random SUB-PIXEL position (a joint does not sit on the 4-pixel grid of the map), plus
    N(0, noise^2) of background.  The top-1 / top-2 margin of such a map is ~ amplitude * (1 - 2 |offset|) / (2 sigma)^2 per axis:
    near-ties between the two pixels that straddle the centre are a property of the maps, not of the arithmetic."""
    cy = 4.0 + uniform01(seed, "peak.cy", (N, K)) * (h - 8.0)
    cx = 4.0 + uniform01(seed, "peak.cx", (N, K)) * (w - 8.0)
    amp = 0.7 + 0.3 * uniform01(seed, "peak.amp", (N, K))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    d2 = (yy[None, None] - cy[..., None, None]) ** 2 + (xx[None, None] - cx[..., None, None]) ** 2
    hm = amp[..., None, None] * np.exp(-d2 / (2.0 * sigma * sigma))
    return torch.from_numpy(hm.astype(np.float32)) + noise * normal(seed, "peak.noise", (N, K, h, w))


# ---- weights ---------------------------------------------------------------------------------------
def _fan_in(name: str, shape: Tuple[int, ...], transposed: bool) -> int:
    if transposed:  # [Cin, Cout, 4, 4]; every output pixel sees 2x2 of the 4x4 taps
        return shape[0] * 4
    return shape[1] * shape[2] * shape[3]


def fill_pose_state_dict(sd: Dict[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """He-scaled conv / deconv / heatmap weights and randomised BN statistics for a
    `deconv('resnetNN', K)` state_dict (keys as SURVEY Appendix B). bn3 / downsample gammas are kept
    small so 16-33 residual blocks do not blow the activations out of fp16 range."""
    out = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(0, dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = normal(seed, k, shape, std=0.1)
        elif k.endswith("running_var"):
            out[k] = uniform(seed, k, shape, 0.5, 1.5)
        elif v.dim() == 4:
            transposed = k.startswith("deconv.")
            gain = 1.0 if k.startswith("heatmap") else 2.0
            std = float(np.sqrt(gain / _fan_in(k, shape, transposed)))
            out[k] = normal(seed, k, shape, std=std)
        elif k.endswith(".weight"):  # BN gamma
            last = ".bn3." in k or ".downsample.1." in k
            out[k] = uniform(seed, k, shape, 0.2, 0.4) if last else uniform(seed, k, shape, 0.5, 1.5)
        elif k.endswith(".bias"):
            out[k] = normal(seed, k, shape, std=0.1)
        else:
            raise KeyError(k)
    return out


def fill_flow_state_dict(sd: Dict[str, torch.Tensor], seed: int) -> Dict[str, torch.Tensor]:
    """He-scaled (LeakyReLU 0.1) conv / deconv weights and small biases for a FlowNet2* state_dict."""
    out = {}
    for k, v in sd.items():
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(0, dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = normal(seed, k, shape, std=0.1)
        elif k.endswith("running_var"):
            out[k] = uniform(seed, k, shape, 0.5, 1.5)
        elif v.dim() == 4:
            leaf = k.split(".")[-2] if k.split(".")[-2] != "0" else k.split(".")[-3]
            transposed = leaf.startswith("deconv") or leaf.startswith("upsampled_flow")
            gain = 1.0 if (leaf.startswith("predict_flow") or leaf.startswith("upsampled_flow")) else 2.0 / 1.01
            std = float(np.sqrt(gain / _fan_in(k, shape, transposed)))
            out[k] = normal(seed, k, shape, std=std)
        elif k.endswith(".bias"):
            out[k] = uniform(seed, k, shape, -0.05, 0.05)
        elif k.endswith(".weight"):  # BN gamma (batchNorm=True variants)
            out[k] = uniform(seed, k, shape, 0.5, 1.5)
        else:
            raise KeyError(k)
    return out
