"""Recipe: can the heat-map conv ALONE be fitted so that a random-weight trunk emits single-peak maps?  

The exact-arg-max mode (DeconvResnet.forward_keypoint_rows_exact) is designed for the heat maps of a TRAINED pose net — one Gaussian
peak per joint — and no trained weights exist in this environment.  The proposal: keep the synthetic trunk, fit only the 256 -> 17
1x1 conv by ridge regression (closed form, on the CPU oracle's deconv features) to the training targets of the reference
(lib/pose/utils/heatmap.py:19-60: exp(-d^2 / (2 sigma^2)) around the joint, sigma = 2 map pixels, tools/pose/config.py:85) for crops
that carry 17 distinguishable markers.  `ridge_fit_report()` does exactly that and measures what comes out on held-out crops.

Result (this file's test pins it): the fit FAILS — validation maps peak at ~0.03 where the target is 1.0, their arg-max lies a
median ~20 map pixels from the marker, the strongest competing peak is 0.95 of the maximum.  A randomly initialised trunk squeezes
the crop through an 8 x 6 x 2048 bottleneck and three random 4x4 deconvs; where in a 64 x 48 map a marker sits is not a LINEAR
function of the 256 features a pixel ends up with.  Peaky maps need trained deconvs, not a trained read-out.  What the mode costs
on such maps is therefore measured in two halves (bench.py: trained_like_record): the screen on synthetic single-peak maps
(synth.peaked_heatmaps) and the mode at the re-run fraction the screen reports for them.

Test infrastructure: imports the oracle (allowed under tests/ only)."""
import math

import numpy as np
import torch

from flowtrack.pytorch_amd import synth

K, H, W = 17, 256, 192


def marker_colours() -> np.ndarray:
    """17 well-spread directions in RGB space (Fibonacci sphere), amplitude 3: crops are ~N(0, 1) after normalisation."""
    i = np.arange(K) + 0.5
    phi = np.arccos(1 - 2 * i / K)
    th = np.pi * (1 + 5 ** 0.5) * i
    return 3.0 * np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)


def marker_crops(seed: int, B: int, sigma_img: float = 5.0, noise: float = 0.25):
    """([B,3,H,W] crops: faint noise + one coloured Gaussian blob per joint, joints [B,K,2] (x, y) in crop pixels)."""
    u = synth.uniform01(seed, "peaky.pts", (B, K, 2))
    pts = np.stack([16 + u[..., 0] * (W - 32), 16 + u[..., 1] * (H - 32)], -1)
    x = noise * synth.normal(seed, "peaky.bg", (B, 3, H, W)).numpy().astype(np.float64)
    yy, xx = np.mgrid[0:H, 0:W]
    col = marker_colours()
    for b in range(B):
        for k in range(K):
            g = np.exp(-((xx - pts[b, k, 0]) ** 2 + (yy - pts[b, k, 1]) ** 2) / (2 * sigma_img ** 2))
            x[b] += col[k][:, None, None] * g[None]
    return torch.from_numpy(x.astype(np.float32)), pts


def gaussian_targets(pts: np.ndarray, h: int = 64, w: int = 48, sigma: float = 2.0) -> np.ndarray:
    """draw_gaussian of the reference (heatmap.py:19-60) for joints given in crop pixels: peak 1 at pts / 4, window +-ceil(3 sigma)."""
    B = pts.shape[0]
    T = np.zeros((B, K, h, w), np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    thr = math.ceil(3 * sigma)
    for b in range(B):
        for k in range(K):
            px, py = pts[b, k] / 4.0
            g = np.exp(-((xx - px) ** 2 + (yy - py) ** 2) / (2 * sigma ** 2))
            g[(np.abs(xx - px) > thr) | (np.abs(yy - py) > thr)] = 0
            T[b, k] = g
    return T


def ridge_fit(features: torch.Tensor, targets: torch.Tensor, lam: float = 1e-2):
    """Closed-form ridge regression of the 1x1 heat-map conv: features [n,256,h,w], targets [n,K,h,w] -> (weight [K,256,1,1], bias [K])."""
    F = features.double().permute(0, 2, 3, 1).reshape(-1, features.shape[1])
    T = targets.double().permute(0, 2, 3, 1).reshape(-1, targets.shape[1])
    A = torch.cat([F, torch.ones(F.shape[0], 1, dtype=torch.float64)], 1)
    G = A.T @ A + lam * A.shape[0] * torch.eye(A.shape[1], dtype=torch.float64)
    Wb = torch.linalg.solve(G, A.T @ T)
    return Wb[:-1].T.reshape(targets.shape[1], features.shape[1], 1, 1).float().contiguous(), Wb[-1].float().contiguous()


def ridge_fit_report(n_fit: int = 16, n_val: int = 8, seed: int = 7):
    """Fit on n_fit marker crops, evaluate on n_val others (all through the CPU oracle, fp32).  Returns the measured quality."""
    from flowtrack.pytorch_amd.pose import models
    from oracle import pose_ref
    m = models.deconv("resnet50", num_classes=K, pretrained=False)
    sd = synth.fill_pose_state_dict(m.state_dict(), seed)
    x, pts = marker_crops(11, n_fit + n_val)
    _, feats = pose_ref.pose_forward(sd, x, return_features=True)
    F = feats["deconv"]
    T = torch.from_numpy(gaussian_targets(pts))
    w, b = ridge_fit(F[:n_fit], T[:n_fit])
    P = torch.nn.functional.conv2d(F[n_fit:], w, b)
    idx = P.flatten(2).argmax(2)
    py, px = (idx // 48).numpy(), (idx % 48).numpy()
    d = np.sqrt((px - pts[n_fit:, :, 0] / 4) ** 2 + (py - pts[n_fit:, :, 1] / 4) ** 2)
    yy, xx = np.mgrid[0:64, 0:48]
    second = []
    for bi in range(n_val):
        for k in range(K):
            away = ((xx - px[bi, k]) ** 2 + (yy - py[bi, k]) ** 2) > 16           # outside 4 pixels of the maximum
            second.append(float(P[bi, k].numpy()[away].max() / P[bi, k].max().item()))
    return {"val_peak_mean": float(P.flatten(2).max(2).values.mean()), "median_argmax_error_map_px": float(np.median(d)),
            "frac_within_2px": float((d < 2).mean()), "second_peak_over_peak_median": float(np.median(second)),
            "fit_crops": n_fit, "val_crops": n_val}


if __name__ == "__main__":
    import json
    torch.set_num_threads(32)
    print(json.dumps(ridge_fit_report()))
