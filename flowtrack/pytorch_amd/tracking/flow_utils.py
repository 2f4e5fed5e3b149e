"""Box propagation by optical flow and box NMS — host side of the tracking glue.

Restates the INTENDED semantics of lib/tracking/flow_utils.py:7-35 (which is broken as written: it takes
H, W from the wrong axes of the [2,H,W] flow (:19-20) and indexes with a list-of-lists that numpy >= 1.23
rejects (:23-24)) and of the CPU box NMS the glue calls (lib/detection/nms/src/nms.c:4-68 via pth_nms.py:5-22,
lib/tracking/net_utils.py:31: `IoU >= thresh` suppresses, pixel-inclusive +1 widths, score-descending order).
"""
from __future__ import annotations

import numpy as np

EXTEND_FACTOR = 0.15  # flow_utils.py:18


def box_propagation(keypoints: np.ndarray, flow: np.ndarray) -> np.ndarray:
    """keypoints [N,K,3] (x, y, score) of the previous frame, flow [2,H,W] (u, v) previous -> current.
    Each keypoint moves by the flow at its (truncated, clipped) pixel; the box of a person is the min/max over
    its keypoints with score > 0, grown by 15 % (7.5 % per side) and clipped to the image. Returns [N,4]."""
    keypoints = np.asarray(keypoints, dtype=np.float64)
    _, H, W = flow.shape
    xs = np.clip(keypoints[..., 0].astype(int), 0, W - 1)
    ys = np.clip(keypoints[..., 1].astype(int), 0, H - 1)
    offset = np.stack((flow[0, ys, xs], flow[1, ys, xs]), axis=-1)        # [N,K,2]
    shifted = keypoints[..., :2] + offset
    mask = (keypoints[..., 2] > 0)[..., None]
    big = float(max(H, W))
    min_ = np.min(np.where(mask, shifted, big), axis=1)                   # flow_utils.py:28
    max_ = np.max(np.where(mask, shifted, 0.0), axis=1)                   # flow_utils.py:29
    extend = (max_ - min_) * EXTEND_FACTOR / 2
    up_left = np.fmax(min_ - extend, 0)
    bottom_right = np.fmin(max_ + extend, np.array([W - 1, H - 1], dtype=np.float64))
    return np.concatenate((up_left, bottom_right), axis=1)


def nms(dets: np.ndarray, thresh: float) -> np.ndarray:
    """dets [n,5] (x1,y1,x2,y2,score) -> indices kept, highest score first (nms.c:36-63, `>=`)."""
    dets = np.asarray(dets, dtype=np.float32)
    if dets.shape[0] == 0:
        return np.zeros((0,), dtype=np.int64)
    x1, y1, x2, y2, scores = dets.T
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = np.argsort(-scores, kind="stable")
    # all pairwise overlaps at once, in score order (the same float32 operations nms.c:45-59 does pair by pair; a box that
    # is already suppressed suppresses nothing because the greedy walk below skips its row), then the walk on plain lists:
    # this function sits on the critical path of the clip's sequential pass, one call per frame on ~10 boxes
    x1, y1, x2, y2, areas = x1[order], y1[order], x2[order], y2[order], areas[order]
    w = np.maximum(0.0, np.minimum(x2[:, None], x2[None, :]) - np.maximum(x1[:, None], x1[None, :]) + 1)
    h = np.maximum(0.0, np.minimum(y2[:, None], y2[None, :]) - np.maximum(y1[:, None], y1[None, :]) + 1)
    inter = w * h
    hit = (inter / (areas[:, None] + areas[None, :] - inter) >= thresh).tolist()
    n = len(hit)
    suppressed = [False] * n
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        row = hit[i]
        for j in range(i + 1, n):
            if row[j]:
                suppressed[j] = True
    return order[keep].astype(np.int64) if keep else np.zeros((0,), dtype=np.int64)
