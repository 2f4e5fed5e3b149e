"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of lib/pose/utils/evaluation.py max_preds (:11-20) and final_preds (:22-35) with
the torch-0.4 semantics the reference was written for: y = idx // W (integer floor — torch >= 1.5
true-divides, SURVEY §8(c) drift item 1) and first-occurrence arg-max on ties.
Pinned against the imported reference (with that one drift corrected) by tests/golden/make_golden.py.
"""
from __future__ import annotations

import numpy as np


def max_preds_ref(heatmap: np.ndarray):
    n, c, h, w = heatmap.shape
    flat = heatmap.reshape(n, c, -1)
    idx = flat.argmax(-1)                      # first occurrence
    scores = np.take_along_axis(flat, idx[..., None], -1)
    coords = np.stack((idx % w, idx // w), axis=2).astype(np.float32)
    coords = coords * (scores > 0).astype(np.float32)
    return coords, scores.astype(np.float32), idx.astype(np.int64)


def _transform(center, scale, res):
    t = np.eye(3)
    t[0, 0] = res[0] / scale
    t[1, 1] = res[0] / scale
    t[0, 2] = -res[0] * center[0] / scale + 0.5 * res[1]
    t[1, 2] = -res[0] * center[1] / scale + 0.5 * res[0]
    return t


def final_preds_ref(heatmap: np.ndarray, center, scale, adjust_coords=False):
    coords, scores, idx = max_preds_ref(heatmap)
    n, c, h, w = heatmap.shape
    if adjust_coords:
        for i in range(n):
            for j in range(c):
                hm = heatmap[i, j]
                x, y = int(coords[i, j, 0]), int(coords[i, j, 1])
                if 0 < x < w - 1 and 0 < y < h - 1:
                    diff = np.array([hm[y, x + 1] - hm[y, x - 1], hm[y + 1, x] - hm[y - 1, x]])
                    coords[i, j] += np.sign(diff) * 0.25
    pre = coords.copy()
    for i in range(n):
        t_inv = np.linalg.inv(_transform(center[i], scale[i], (h, w)))
        pts = np.concatenate((coords[i], np.ones((c, 1))), axis=1)
        coords[i] = np.dot(t_inv, pts.T)[:2].T
    return coords, scores, idx, pre
