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
        boxes = np.asarray(boxes, dtype=np.float64).reshape(len(keypoints), -1)
        ids = list(self.tracks)
        if flow is not None and ids:
            moved = propagate_keypoints(np.stack([self.tracks[i]["kpts"] for i in ids]), flow)
        else:
            moved = np.stack([self.tracks[i]["kpts"] for i in ids]) if ids else np.zeros((0,) + keypoints.shape[1:])
        scores = boxes[:, 4] if boxes.shape[1] > 4 else keypoints[..., 2].mean(1)
        order = np.argsort(-scores, kind="stable")
        taken, assigned = set(), [-1] * len(keypoints)
        for d in order:
            area = max((boxes[d, 2] - boxes[d, 0]) * (boxes[d, 3] - boxes[d, 1]), 1.0)
            best, best_s = None, self.oks_threshold
            for ti, tid in enumerate(ids):
                if tid in taken:
                    continue
                s = pose_oks(keypoints[d], moved[ti], area, kpt_thresh=self.kpt_threshold)
                if s > best_s:
                    best, best_s = tid, s
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
