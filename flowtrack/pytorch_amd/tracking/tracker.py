"""Flow-based pose tracking — absent from the reference (README.md:14 "Pose tracking" is
unchecked; tools/tracking/demo.py stops after loading the nets).  Written from the method the reference
implements pieces of ("Simple Baselines for Human Pose Estimation and Tracking", Xiao et al., §3.2-3.3):
  * joint propagation: every tracked pose of frame t-1 is moved to frame t by the optical flow at its joints;
  * flow-based pose similarity: OKS between a detected pose of frame t and the propagated poses
    (building block: compute_oks, lib/pose/utils/evaluation.py:61-82);
  * greedy matching: detections in descending score order take the best still-unmatched track with
    similarity above a threshold, otherwise they start a new id.
Host-side Python (north_star keeps the tracking logic on the host).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from ..pose.evaluation import COCO_DELTA


def propagate_keypoints(keypoints: np.ndarray, flow: np.ndarray) -> np.ndarray:
    """[N,K,3] keypoints of frame t-1 + flow [2,H,W] (t-1 -> t) -> keypoints moved into frame t."""
    kp = np.array(keypoints, dtype=np.float64, copy=True)
    _, H, W = flow.shape
    xs = np.clip(kp[..., 0].astype(int), 0, W - 1)
    ys = np.clip(kp[..., 1].astype(int), 0, H - 1)
    kp[..., 0] += flow[0, ys, xs]
    kp[..., 1] += flow[1, ys, xs]
    return kp


def pose_oks(a: np.ndarray, b: np.ndarray, area: float, delta=COCO_DELTA, kpt_thresh: float = 0.0) -> float:
    """OKS between two poses [K,3] over joints both score above kpt_thresh (evaluation.py:72-78)."""
    counted = np.logical_and(a[:, 2] > kpt_thresh, b[:, 2] > kpt_thresh)
    if not counted.any():
        return 0.0
    d2 = ((a[counted, :2] - b[counted, :2]) ** 2).sum(1)
    return float(np.exp(-d2 / 2 / (delta[counted] ** 2) / (area + np.spacing(1))).mean())


def pose_oks_matrix(dets: np.ndarray, tracks: np.ndarray, areas: np.ndarray, delta=COCO_DELTA,
                    kpt_thresh: float = 0.0) -> np.ndarray:
    """pose_oks of every (detection, track) pair at once: dets [D,K,3], tracks [T,K,3], areas [D] -> [D,T]."""
    if len(dets) == 0 or len(tracks) == 0:
        return np.zeros((len(dets), len(tracks)))
    diff = dets[:, None, :, :2] - tracks[None, :, :, :2]
    d2 = np.einsum("dtkc,dtkc->dtk", diff, diff)
    e = np.exp(-d2 / 2 / (np.asarray(delta) ** 2)[None, None, :] / (areas[:, None, None] + np.spacing(1)))
    if dets[..., 2].min() > kpt_thresh and tracks[..., 2].min() > kpt_thresh:
        return e.mean(-1)            # every joint of every pair counts (the usual case: one reduction instead of five)
    counted = np.logical_and(dets[:, None, :, 2] > kpt_thresh, tracks[None, :, :, 2] > kpt_thresh)       # [D,T,K]
    n = counted.sum(-1)
    return np.where(n > 0, (e * counted).sum(-1) / np.maximum(n, 1), 0.0)


@dataclass
class FlowTracker:
    oks_threshold: float = 0.5
    kpt_threshold: float = 0.0
    max_age: int = 1                       # frames a track survives without a match
    next_id: int = 0
    tracks: Dict[int, dict] = field(default_factory=dict)   # id -> {"kpts": [K,3], "age": int}

    def update(self, keypoints: np.ndarray, boxes: np.ndarray, flow: np.ndarray = None) -> List[int]:
        """keypoints [N,K,3], boxes [N,>=4] (x1,y1,x2,y2[,score]) of the current frame, flow from the previous
        frame (None on the first frame). Returns the track id of every detection."""
        keypoints = np.asarray(keypoints, dtype=np.float64)
        boxes = np.asarray(boxes, dtype=np.float64)
        if len(keypoints) == 0:        # a frame without detections: nothing to match, tracks still move and age
            keypoints = keypoints.reshape((0,) + (keypoints.shape[1:] if keypoints.ndim == 3 else (len(COCO_DELTA), 3)))
            boxes = np.zeros((0, boxes.shape[-1] if boxes.ndim == 2 and boxes.shape[-1] >= 4 else 5))
        else:
            boxes = boxes.reshape(len(keypoints), -1)
        ids = list(self.tracks)
        if flow is not None and ids:
            moved = propagate_keypoints(np.stack([self.tracks[i]["kpts"] for i in ids]), flow)
        else:
            moved = np.stack([self.tracks[i]["kpts"] for i in ids]) if ids else np.zeros((0,) + keypoints.shape[1:])
        scores = boxes[:, 4] if boxes.shape[1] > 4 else keypoints[..., 2].mean(1)
        order = np.argsort(-scores, kind="stable")
        taken, assigned = set(), [-1] * len(keypoints)
        areas = np.maximum((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), 1.0) if len(boxes) else np.zeros(0)
        sim = pose_oks_matrix(keypoints, moved, areas, kpt_thresh=self.kpt_threshold)     # [detections, tracks]
        free = np.ones(len(ids), dtype=bool)
        for d in order:
            best = None
            if free.any():
                row = np.where(free, sim[d], -1.0)
                ti = int(np.argmax(row))                  # first maximum = the loop's "first strictly better" track
                if row[ti] > self.oks_threshold:
                    best = ids[ti]
                    free[ti] = False
            if best is None:
                best = self.next_id
                self.next_id += 1
            taken.add(best)
            assigned[d] = best
        for tid in ids:                         # age out unmatched tracks, keep their propagated pose
            if tid not in taken:
                self.tracks[tid]["age"] += 1
                self.tracks[tid]["kpts"] = moved[ids.index(tid)]
                if self.tracks[tid]["age"] > self.max_age:
                    del self.tracks[tid]
        for d, tid in enumerate(assigned):
            self.tracks[tid] = {"kpts": keypoints[d].copy(), "age": 0}
        return assigned
