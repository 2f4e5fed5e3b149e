"""Heatmaps -> keypoints and the OKS metrics, host side.

Mirrors lib/pose/utils/evaluation.py (max_preds :11-20, final_preds :22-35, compute_oks :61-82,
nms_oks :84-101, eval_mAP :190-211; nms_heatmap :37-59, calc_dists / dist_acc / accuracy :104-159 and compute_pck :164-175
at the end of the file) and the point transform of lib/pose/utils/transforms.py
(:173-226, rot = 0).  The per-map arg-max and the +/-0.25 px nudge — a Python double loop that
indexes a GPU tensor element by element in the reference — run as one HIP launch
(ft_heatmap_max_preds); the 3x3 inverse affine stays numpy float64 as in the reference.

torch-0.4 semantics are kept where torch 2.x drifted: y = idx // W (integer floor).
"""
from __future__ import annotations

import numpy as np
import torch

from ..hip_ops import heatmap_max_preds

# COCO per-keypoint constants (lib/pose/datasets/coco.py:233)
COCO_DELTA = 2 * np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def max_preds(heatmap: torch.Tensor, threshold: float = 0):
    """[N,K,H,W] -> (coords [N,K,2] (x,y) fp32 numpy, scores [N,K,1] fp32 numpy); coords are zeroed
    where score <= threshold (evaluation.py:17-19).  Only threshold 0 runs on the device path."""
    if threshold != 0:
        raise ValueError("the HIP max_preds implements the reference default threshold=0")
    _, scores, coords = heatmap_max_preds(heatmap, adjust_coords=False)
    return coords.cpu().numpy(), scores.cpu().numpy()


def get_transform(center, scale, res):
    """Affine image->heatmap map for rot=0, factor=1 (transforms.py:173-184). res = (h, w)."""
    t = np.eye(3)
    t[0, 0] = res[0] / scale
    t[1, 1] = res[0] / scale
    t[0, 2] = -res[0] * center[0] / scale + 0.5 * res[1]
    t[1, 2] = -res[0] * center[1] / scale + 0.5 * res[0]
    return t


def transform_preds(coords: np.ndarray, center, scale, res) -> np.ndarray:
    """Heatmap pixels -> image pixels with the inverse crop affine, in place per sample
    (transforms.py:221-226)."""
    for i in range(coords.shape[0]):
        t_inv = np.linalg.inv(get_transform(center[i], scale[i], res))
        pts = np.concatenate((coords[i], np.ones((coords[i].shape[0], 1))), axis=1)
        coords[i] = np.dot(t_inv, pts.T)[:2].T
    return coords


def final_preds(heatmap: torch.Tensor, center, scale, adjust_coords: bool = False):
    """(coords in image pixels [N,K,2], scores [N,K,1]) as evaluation.py:22-35."""
    n, c, h, w = heatmap.shape
    _, scores, coords = heatmap_max_preds(heatmap, adjust_coords=adjust_coords)
    coords = transform_preds(coords.cpu().numpy(), np.asarray(center, dtype=np.float64),
                             np.asarray(scale, dtype=np.float64), (h, w))
    return coords, scores.cpu().numpy()


def compute_oks(pred, anno, ref_scale, delta, ground_truth: bool = True, threshold: float = 0):
    """Object keypoint similarity per sample (evaluation.py:61-82)."""
    pred = np.asarray(pred)
    anno = np.asarray(anno)
    n = pred.shape[0]
    if anno.ndim < 3:
        anno = np.tile(anno[np.newaxis], (n, 1, 1))
    oks = np.zeros(n)
    for i in range(n):
        if ground_truth:
            counted = anno[i][:, 2] > 0
        else:
            counted = np.logical_and(anno[i][:, 2] >= threshold, pred[i][:, 2] >= threshold)
        if counted.sum() != 0:
            d2 = ((anno[i][counted, :2] - pred[i][counted, :2]) ** 2).sum(1)
            k2 = delta[counted] ** 2
            oks[i] = np.exp(-d2 / 2 / k2 / (ref_scale[i] + np.spacing(1))).mean()
    return oks


def nms_oks(pred, oks_thresh, delta, kpt_thresh: float = 0):
    """Greedy OKS-NMS over dicts with 'score', 'joints', 'area' (evaluation.py:84-101)."""
    scores = np.array([p["score"] for p in pred])
    joints = np.array([p["joints"] for p in pred])
    areas = np.array([p["area"] for p in pred])
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        rest = order[1:]
        ref = (areas[i] + areas[rest]) / 2
        oks = compute_oks(joints[rest], joints[i], ref, delta, ground_truth=False, threshold=kpt_thresh)
        order = rest[np.where(oks <= oks_thresh)[0]]
    return keep


def eval_mAP(pred, anno, ref_scale, delta):
    """Fraction of samples with OKS above each of the 10 COCO thresholds (evaluation.py:190-211)."""
    all_oks = np.concatenate([compute_oks(pred[i], anno[i], ref_scale[i], delta) for i in range(len(pred))])
    return [float(np.sum(all_oks > thr) / np.float32(all_oks.size)) for thr in np.linspace(0.5, 0.95, 10)]


# ---- the remaining host-side helpers of lib/pose/utils/evaluation.py (training-time PCK read-out, peak NMS) -----------
# Not on the inference hot path: they run as plain torch / numpy on whatever device the maps live on, as in the reference.
def _argmax_xy(heatmap: torch.Tensor):
    """Per-map arg-max -> (x, y, score) with torch-0.4 integer semantics (y = idx // W, first occurrence wins)."""
    n, c, h, w = heatmap.shape
    scores, idx = torch.max(heatmap.reshape(n, c, -1), -1)
    x = torch.remainder(idx, w).float()
    y = torch.div(idx, w, rounding_mode="floor").float()
    return x, y, scores


def nms_heatmap(heatmap: torch.Tensor, threshold: float = 0, window_size: int = 3) -> np.ndarray:
    """Strongest local maximum of every map -> [N, K, 3] (x, y, score) (evaluation.py:37-59): a pixel survives when it
    equals the max of its window (and is >= threshold when threshold > 0); the rest are zeroed before the arg-max."""
    pad = (window_size - 1) // 2
    max_map = torch.nn.functional.max_pool2d(heatmap, kernel_size=window_size, stride=1, padding=pad)
    mask = torch.eq(heatmap, max_map)
    if threshold > 0:
        mask = mask & heatmap.ge(threshold)
    x, y, scores = _argmax_xy(heatmap * mask.to(heatmap.dtype))
    return torch.stack((x, y, scores), dim=2).cpu().numpy()


def calc_dists(preds: np.ndarray, target: np.ndarray, normalize: np.ndarray) -> np.ndarray:
    """[K, N] normalised distances, -1 where the target joint is not annotated (x or y < 1) (evaluation.py:104-116)."""
    preds = np.asarray(preds, dtype=np.float32)
    target = np.asarray(target, dtype=np.float32)
    norm = np.asarray(normalize)[:, None, :]
    d = np.linalg.norm(preds / norm - target / norm, axis=2).astype(np.float64)
    annotated = (target[..., 0] >= 1) & (target[..., 1] >= 1)
    return np.where(annotated, d, -1.0).T


def dist_acc(dists: np.ndarray, thr: float = 0.5) -> float:
    """Fraction of the counted distances (!= -1) below thr, or -1 when none is counted (evaluation.py:119-126)."""
    counted = np.not_equal(dists, -1)
    num = counted.sum()
    return float(np.less(dists[counted], thr).sum() * 1.0 / num) if num > 0 else -1


def accuracy(output: torch.Tensor, target: torch.Tensor, hm_type: str = "gaussian", thr: float = 0.5):
    """PCK between the arg-max of predicted and ground-truth heat maps (evaluation.py:130-159):
    (acc [K + 1] with the mean in slot 0, mean accuracy, joints counted, predicted coords [N, K, 2])."""
    n, c, h, w = output.shape
    if hm_type != "gaussian":
        raise ValueError("only hm_type='gaussian' is defined by the reference")

    def coords_of(hm):
        x, y, s = _argmax_xy(hm)
        return (torch.stack((x, y), dim=2) * s.gt(0).float().unsqueeze(-1)).cpu().numpy()      # max_preds' threshold-0 mask
    pred, tgt = coords_of(output), coords_of(target)
    norm = np.ones((n, 2)) * np.array([h, w]) / 10
    dists = calc_dists(pred, tgt, norm)
    acc = np.zeros(c + 1)
    total, cnt = 0.0, 0
    for i in range(c):
        acc[i + 1] = dist_acc(dists[i], thr)
        if acc[i + 1] >= 0:
            total += acc[i + 1]
            cnt += 1
    avg = total / cnt if cnt else 0
    if cnt:
        acc[0] = avg
    return acc, avg, cnt, pred


def compute_pck(pred: np.ndarray, anno: np.ndarray, ref_scale, threshold: float) -> np.ndarray:
    """Per-joint PCK(h) (evaluation.py:164-175; the reference's +1 on the predicted joints — MATLAB-indexed annotations —
    is kept)."""
    pred, anno = np.asarray(pred), np.asarray(anno)
    counted = anno[:, :, 2] > 0
    dists = np.linalg.norm(pred[:, :, :2] + 1 - anno[:, :, :2], axis=2) / ref_scale
    match = (dists <= threshold) * counted
    return match.sum(0).astype(np.float32) / counted.sum(0)
