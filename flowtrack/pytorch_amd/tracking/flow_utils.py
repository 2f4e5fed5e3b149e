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
    suppressed = np.zeros(dets.shape[0], dtype=bool)
    keep = []
    for _i in range(order.size):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        rest = rest[~suppressed[rest]]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr >= thresh]] = True
    return np.asarray(keep, dtype=np.int64)
