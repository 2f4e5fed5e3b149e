"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Independent CPU restatements for the tracking glue ("next" rows N1/N2):
  crop_affine_ref      the crop geometry of lib/pose/utils/transforms.py:173-184,231-240 (cv2.warpAffine with
                       t = get_transform(center, scale, res): inverse map, bilinear, constant-0 border), in
                       float64 numpy.  cv2 is absent here and quantises weights to 1/32 px => UNPINNED.
  nms_ref              literal loop restatement of lib/detection/nms/src/nms.c:33-64 (IoU >= thresh, +1 widths)
                       with the score-descending order of pth_nms.py:16.
  box_propagation_ref  intended semantics of lib/tracking/flow_utils.py:7-35, per person / per joint loops
                       (the reference function itself does not run: wrong H/W axes, numpy-2-incompatible index).
"""
from __future__ import annotations

import numpy as np


def crop_affine_ref(img: np.ndarray, center, scale, res, mean=None, inv_std=None, pre_scale=1.0) -> np.ndarray:
    """img [H,W,C] uint8 -> [C,rh,rw] float32."""
    H, W, C = img.shape
    rh, rw = res
    t = np.eye(3)
    t[0, 0] = t[1, 1] = rh / scale
    t[0, 2] = -rh * center[0] / scale + 0.5 * rw
    t[1, 2] = -rh * center[1] / scale + 0.5 * rh
    tinv = np.linalg.inv(t)
    ys, xs = np.meshgrid(np.arange(rh), np.arange(rw), indexing="ij")
    src = tinv @ np.stack((xs.ravel(), ys.ravel(), np.ones(rh * rw)))
    sx, sy = src[0].reshape(rh, rw), src[1].reshape(rh, rw)
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    ax, ay = sx - x0, sy - y0
    f = img.astype(np.float64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok[..., None], f[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)
    out = ((1 - ay) * (1 - ax))[..., None] * tap(y0, x0) + ((1 - ay) * ax)[..., None] * tap(y0, x0 + 1) + \
          (ay * (1 - ax))[..., None] * tap(y0 + 1, x0) + (ay * ax)[..., None] * tap(y0 + 1, x0 + 1)
    out = out * pre_scale
    if mean is not None:
        out = out - np.asarray(mean)
    if inv_std is not None:
        out = out * np.asarray(inv_std)
    return out.transpose(2, 0, 1).astype(np.float32)


def nms_ref(dets: np.ndarray, thresh: float):
    dets = np.asarray(dets, dtype=np.float32)
    n = dets.shape[0]
    areas = (dets[:, 2] - dets[:, 0] + 1) * (dets[:, 3] - dets[:, 1] + 1)
    order = sorted(range(n), key=lambda i: -dets[i, 4])
    suppressed = [False] * n
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        for _j in range(_i + 1, n):
            j = order[_j]
            if suppressed[j]:
                continue
            w = max(0.0, min(dets[i, 2], dets[j, 2]) - max(dets[i, 0], dets[j, 0]) + 1)
            h = max(0.0, min(dets[i, 3], dets[j, 3]) - max(dets[i, 1], dets[j, 1]) + 1)
            inter = w * h
            if inter / (areas[i] + areas[j] - inter) >= thresh:
                suppressed[j] = True
    return keep


def box_propagation_ref(keypoints: np.ndarray, flow: np.ndarray, extend_factor=0.15) -> np.ndarray:
    _, H, W = flow.shape
    boxes = []
    for person in keypoints:
        pts = []
        for (x, y, s) in person:
            xi, yi = min(max(int(x), 0), W - 1), min(max(int(y), 0), H - 1)
            pts.append((x + flow[0, yi, xi], y + flow[1, yi, xi], s))
        vis = [(px, py) for px, py, s in pts if s > 0]
        mn = np.array([min([p[0] for p in vis] + ([] if len(vis) == len(pts) else [max(H, W)])),
                       min([p[1] for p in vis] + ([] if len(vis) == len(pts) else [max(H, W)]))]) if vis else np.array([max(H, W)] * 2, float)
        mx = np.array([max([p[0] for p in vis] + ([] if len(vis) == len(pts) else [0.0])),
                       max([p[1] for p in vis] + ([] if len(vis) == len(pts) else [0.0]))]) if vis else np.zeros(2)
        ext = (mx - mn) * extend_factor / 2
        ul = np.maximum(mn - ext, 0)
        br = np.minimum(mx + ext, [W - 1, H - 1])
        boxes.append(np.concatenate((ul, br)))
    return np.asarray(boxes)
