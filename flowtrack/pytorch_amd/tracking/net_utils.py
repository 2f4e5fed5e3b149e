"""Per-frame glue between the detector boxes, the pose net and the flow net.

Mirrors lib/tracking/net_utils.py (detect :14-34, pose_est :36-71, flow_est :73-92) with its intended
semantics; what the reference gets wrong is fixed, not copied: `end = max(num_boxes, ...)` (:61) is a `min`,
crops are normalised before the net (the reference feeds raw 0..255 BGR, :52-63, while its training pipeline
normalises, lib/pose/datasets/mpii.py:149), the non-existent `transfrom_image` import (:11) is the GPU crop.
The person DETECTOR is out of scope: `detect` takes the detector's person boxes as input
and does the part the glue owns — union with the flow-propagated boxes and box NMS.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _lib
from .._lib import check
from ..hip_ops import current_stream_handle, require_gpu
from ..pose.evaluation import final_preds
from .flow_utils import nms

# BGR ImageNet statistics on the 0..1 scale (lib/pose/datasets/mpii.py:53-54)
BGR_MEAN = (0.406, 0.456, 0.485)
BGR_STD = (0.225, 0.224, 0.229)


def detect(person_dets: np.ndarray, thresh: float = 0.3, prop_dets: np.ndarray = None) -> np.ndarray:
    """Union of detector boxes [n,5] and propagated boxes [m,5], then box NMS (net_utils.py:28-32)."""
    dets = np.asarray(person_dets, dtype=np.float32).reshape(-1, 5)
    if prop_dets is not None and len(prop_dets):
        dets = np.concatenate((dets, np.asarray(prop_dets, dtype=np.float32).reshape(-1, 5)), axis=0)
    return dets[nms(dets, thresh)]


def boxes_to_center_scale(boxes: np.ndarray, inp_res=(256, 192)):
    """Box -> (center, scale) of the pose crop: aspect-corrected box height (net_utils.py:47-48; the datasets
    additionally enlarge the box by 1.25, lib/pose/datasets/coco.py:110-113 — multiply the returned scales by 1.25 for that)."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    centers = np.stack((boxes[:, [0, 2]].mean(1), boxes[:, [1, 3]].mean(1)), axis=1)
    scales = np.maximum(boxes[:, 3] - boxes[:, 1], (boxes[:, 2] - boxes[:, 0]) / inp_res[1] * inp_res[0])
    return centers, scales


def cv2_crop_matrices(centers: np.ndarray, scales: np.ndarray, inp_res=(256, 192), out: np.ndarray = None) -> np.ndarray:
    """Per box the dst -> src 2x3 map `cv2.warpAffine(img, t[:2], ...)` works with, float64 [N,6]: t = get_transform(center,
    scale, res) (lib/pose/utils/transforms.py:173-184, rot = 0) and then the inversion cv::warpAffine performs on the matrix it
    is handed (imgwarp.cpp: D = M0 * M4 - M1 * M3; D = 1 / D; ...), every operation in double and in the same order, so the
    fixed-point coordinates the kernel derives are cv2's to the last bit.  With rot = 0: M1 = M3 = 0."""
    c = np.asarray(centers, dtype=np.float64).reshape(-1, 2)
    sc = np.asarray(scales, dtype=np.float64).reshape(-1)
    h, w = float(inp_res[0]), float(inp_res[1])
    m0 = h / sc                                             # t[0,0] = t[1,1] = res_ / scale
    m2 = -h * c[:, 0] / sc + 0.5 * w                        # t[0,2]
    m5 = -h * c[:, 1] / sc + 0.5 * h                        # t[1,2]
    zero = np.zeros_like(m0)
    D = m0 * m0 - zero * zero
    with np.errstate(divide="ignore", invalid="ignore"):
        D = np.where(D != 0, 1.0 / D, 0.0)
    a11, a22 = m0 * D, m0 * D
    n1, n3 = zero * -D, zero * -D
    M = np.empty((len(sc), 6), dtype=np.float64) if out is None else out
    M[:, 0], M[:, 1], M[:, 3], M[:, 4] = a11, n1, n3, a22
    M[:, 2] = -a11 * m2 - n1 * m5
    M[:, 5] = -n3 * m2 - a22 * m5
    return M


def crop_boxes(frame_dev: torch.Tensor, centers: np.ndarray, scales: np.ndarray, inp_res=(256, 192), normalize=True,
               cv2_exact: bool = False, return_u8: bool = False):
    """frame_dev: uint8 [H,W,3] (BGR) on the GPU -> crops [N,3,h,w] fp32 on the GPU in one launch
    (ft_crop_affine_fwd; replaces N x cv2.warpAffine + N H2D copies, net_utils.py:49-57).
    cv2_exact: the crops are cv2.warpAffine's uint8 values as the restated classic OpenCV path computes them, bit for bit (unpinned
    against a real cv2 build; ft_crop_affine_cv2_fwd: OpenCV's fixed-point
    INTER_LINEAR) before the normalisation; return_u8 additionally returns them as uint8 [N,h,w,3] = cv2's return value."""
    require_gpu(frame_dev.device)
    if frame_dev.dtype != torch.uint8 or frame_dev.dim() != 3 or not frame_dev.is_contiguous():
        raise ValueError("frame must be a contiguous uint8 [H,W,C] device tensor")
    lib = _lib.load()
    H, W, C = frame_dev.shape
    n = len(scales)
    if cv2_exact or return_u8:
        minv = torch.from_numpy(cv2_crop_matrices(centers, scales, inp_res)).to(frame_dev.device)
        out = torch.empty((n, C, inp_res[0], inp_res[1]), dtype=torch.float32, device=frame_dev.device)
        u8 = torch.empty((n, inp_res[0], inp_res[1], C), dtype=torch.uint8, device=frame_dev.device) if return_u8 else None
        mean = inv_std = None
        pre = 1.0
        if normalize:
            mean = torch.tensor(BGR_MEAN[:C], dtype=torch.float32, device=frame_dev.device)
            inv_std = torch.tensor([1.0 / s for s in BGR_STD[:C]], dtype=torch.float32, device=frame_dev.device)
            pre = 1.0 / 255.0
        check(lib.ft_crop_affine_cv2_fwd(frame_dev.data_ptr(), H, W, C, minv.data_ptr(), n, inp_res[0], inp_res[1],
                                         mean.data_ptr() if mean is not None else None,
                                         inv_std.data_ptr() if inv_std is not None else None, pre,
                                         u8.data_ptr() if u8 is not None else None, out.data_ptr(),
                                         current_stream_handle()), "ft_crop_affine_cv2_fwd")
        return (out, u8) if return_u8 else out
    params = torch.from_numpy(np.concatenate((np.asarray(centers, np.float32).reshape(n, 2),
                                              np.asarray(scales, np.float32).reshape(n, 1)), axis=1)).to(frame_dev.device)
    out = torch.empty((n, C, inp_res[0], inp_res[1]), dtype=torch.float32, device=frame_dev.device)
    mean = inv_std = None
    pre = 1.0
    if normalize:
        mean = torch.tensor(BGR_MEAN[:C], dtype=torch.float32, device=frame_dev.device)
        inv_std = torch.tensor([1.0 / s for s in BGR_STD[:C]], dtype=torch.float32, device=frame_dev.device)
        pre = 1.0 / 255.0
    check(lib.ft_crop_affine_fwd(frame_dev.data_ptr(), H, W, C, params.data_ptr(), n, inp_res[0], inp_res[1],
                                 mean.data_ptr() if mean is not None else None,
                                 inv_std.data_ptr() if inv_std is not None else None, pre, out.data_ptr(),
                                 current_stream_handle()), "ft_crop_affine_fwd")
    return out


def pose_est(net, frame_dev: torch.Tensor, boxes: np.ndarray, inp_res=(256, 192), max_batch=32, normalize=True):
    """Single-person pose for every box of a frame -> keypoints [N,K,3] (x, y, score) in image pixels."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0, _num_joints(net), 3), dtype=np.float32)
    centers, scales = boxes_to_center_scale(boxes, inp_res)
    crops = crop_boxes(frame_dev, centers, scales, inp_res, normalize)
    out_c, out_s = [], []
    for lo in range(0, n, max_batch):
        hi = min(n, lo + max_batch)                         # the reference's max() here is a bug (net_utils.py:61)
        m = hi - lo
        bucket = next(b for b in (4, 8, 16, 32, 64, 128, 1 << 30) if b >= m or b >= max_batch)
        bucket = min(bucket, max(max_batch, m))             # one cached plan (HIP graph) per bucket, not per count
        batch = crops[lo:hi]
        if bucket > m:
            batch = torch.cat((batch, batch.new_zeros((bucket - m,) + tuple(batch.shape[1:]))), 0)
        hm = net(batch)[:m]
        c, s = final_preds(hm, centers[lo:hi], scales[lo:hi], adjust_coords=True)
        out_c.append(c)
        out_s.append(s)
    return np.concatenate((np.concatenate(out_c), np.concatenate(out_s)), axis=2).astype(np.float32)


def _num_joints(net) -> int:
    """Key points per person of `net` (17 COCO / 16 MPII, tools/pose/main.py:22,57): the model API's num_classes."""
    return int(getattr(net, "num_classes", 17))


def heatmap_rows_to_image(rows: np.ndarray, centers: np.ndarray, scales: np.ndarray, hm_hw) -> np.ndarray:
    """Key-point rows (x, y, score) in heat-map pixels -> image pixels: the inverse of the crop affine of
    lib/pose/utils/transforms.py:173-184 (rot = 0) in closed form, for all boxes at once — t = [[a, 0, bx], [0, a, by]] with
    a = h / scale, bx = -h * cx / scale + w / 2, by = -h * cy / scale + h / 2, so x_img = (x - w / 2) * scale / h + cx
    (evaluation.transform_preds does the same through one 3x3 numpy inverse per box)."""
    h, w = hm_hw
    out = np.asarray(rows, dtype=np.float64).copy()
    k = (np.asarray(scales, dtype=np.float64) / h)[:, None]
    c = np.asarray(centers, dtype=np.float64)
    out[..., 0] = (out[..., 0] - 0.5 * w) * k + c[:, 0:1]
    out[..., 1] = (out[..., 1] - 0.5 * h) * k + c[:, 1:2]
    return out.astype(np.float32)


class PoseRunner:
    """Pose of a few boxes of one frame with ONE device round trip, asynchronously (the sequential pass of the clip pipeline,
    lib/tracking/net_utils.py:36-71 + tools/tracking/demo.py:35-42): box parameters go up through a pinned buffer, the crop
    kernel writes straight into the pose plan's input (zero-copy), the plan's graph ends in the key-point rows launch
    (arg-max + 0.25 px nudge on the device), the [bucket, K, 3] rows come back into a pinned buffer behind an event.
    `submit` returns at once; `result` waits for the event — whatever the host does in between overlaps the GPU.
    One slot (parameter + row buffers) per bucket: a submit that finds its bucket's slot still in flight (no result() yet)
    first waits for that launch's event — the crop kernel reads the box parameters straight from the slot's pinned buffer and
    the plan's input is shared, so rewriting either under a pending launch would corrupt it (the tracking pass never does:
    frame t + 1's boxes depend on frame t's rows).  More boxes than the largest bucket are chunked.  The host side of a submit is on the critical
    path of that pass (the GPU idles while boxes are prepared), hence the numpy views of the pinned buffers and the single
    plan look-up per call (tools/dev/clip_profile.py: 0.21 -> ~0.1 ms per frame)."""
    BUCKETS = (4, 8, 16, 32, 64, 128, 256)

    def __init__(self, net, inp_res=(256, 192), normalize=True, replica: int = 0, stream=None, cv2_exact: bool = False):
        """replica / stream: a runner of its own plan copies (DeconvResnet.plan_for(..., replica)) whose launches go to
        `stream` — several runners on one net then overlap on the GPU (one per clip, tools/tracking/demo.run_clips).
        cv2_exact: crops in cv2.warpAffine's fixed-point arithmetic (bit-exact to the restated classic OpenCV path, unpinned against a
        real cv2 build: ft_crop_affine_cv2_fwd) instead of the fp32 bilinear."""
        self.net, self.inp_res = net, inp_res
        self.replica, self.stream = int(replica), stream
        self.cv2_exact = bool(cv2_exact)
        self.K = _num_joints(net)
        self.dev = next(net.parameters()).device
        require_gpu(self.dev)
        self.lib = _lib.load()
        net.keypoints_in_plan = True                      # rows with the adjust_coords nudge, inside the plan's graph
        self.mean = self.inv_std = None
        self.pre = 1.0
        if normalize:
            self.mean = torch.tensor(BGR_MEAN, dtype=torch.float32, device=self.dev)
            self.inv_std = torch.tensor([1.0 / v for v in BGR_STD], dtype=torch.float32, device=self.dev)
            self.pre = 1.0 / 255.0
        self.mean_ptr = self.mean.data_ptr() if self.mean is not None else None
        self.inv_std_ptr = self.inv_std.data_ptr() if self.inv_std is not None else None
        self.slots = {}
        self._plans = {}                                   # bucket -> (plan, the net's plan dict it came from)
        self._calls = 0
        self._checks = None

    def _stream_handle(self):
        """The stream the runner's launches go to, as the C ABI takes it: its own, else torch's current one."""
        if self.stream is not None:
            return ctypes.c_void_p(self.stream.cuda_stream)
        return current_stream_handle(self.dev)

    def _slot(self, bucket: int):
        sl = self.slots.get(bucket)
        if sl is None:
            # box parameters live in pinned host memory the crop kernel reads directly (host allocations are mapped into the
            # device's address space at the same address): no H2D copy op on the stream, no staging tensor
            ph = (torch.zeros((bucket, 6), dtype=torch.float64) if self.cv2_exact else torch.zeros((bucket, 3), dtype=torch.float32)).pin_memory()
            rh = torch.zeros((bucket, self.K, 3), dtype=torch.float32).pin_memory()
            ev = ctypes.c_void_p()
            check(self.lib.ft_event_create(ctypes.byref(ev)), "ft_event_create")
            sl = self.slots[bucket] = {"params_host": ph, "params_np": ph.numpy(), "params_ptr": ph.data_ptr(), "rows_host": rh,
                                       "rows_np": rh.numpy(), "rows_ptr": rh.data_ptr(), "rows_bytes": rh.numel() * 4, "event": ev,
                                       "pending": False}
        elif sl["pending"]:
            check(self.lib.ft_event_synchronize(sl["event"]))   # slot-busy guard: the launch that reads this slot has not finished
            sl["pending"] = False
        return sl

    def _fill_params(self, sl, centers, scales, n, bucket):
        pn = sl["params_np"]
        if self.cv2_exact:
            cv2_crop_matrices(centers, scales, self.inp_res, out=pn[:n])
        else:
            pn[:n, :2] = centers
            pn[:n, 2] = scales
        if n < bucket:
            pn[n:] = pn[0]                                 # padding crops repeat box 0 (their rows are dropped)

    def _plan(self, bucket: int):
        """The bucket's plan without the model's per-call bookkeeping: DeconvResnet.plan_for() re-derives device / dtype and
        sums the version counters of every parameter (~25 us; the whole host side of a submit is ~100 us and sits on the critical
        path of the tracking pass).  The runner keeps the plan and goes back to plan_for() when the model dropped its plans
        (load_state_dict / .to() / refresh() / an in-place parameter edit).  The version-counter check itself
        (HipModule._check_fingerprint, ~20 us) and the eval-mode check run on EVERY submit (skipping them let up to
        31 submits replay graphs built from edited weights); a changed fingerprint swaps net._plans, which the identity test
        below sees."""
        self._calls += 1
        if self._checks is None:                            # (stand-in nets of the tests implement plan_for / replay only)
            self._checks = tuple(f for f in (getattr(self.net, "_check_eval", None), getattr(self.net, "_check_fingerprint", None)) if f)
        for f in self._checks:
            f()
        ent = self._plans.get(bucket)
        if ent is None or ent[1] is not self.net._plans:
            plan = self.net.plan_for(bucket, self.inp_res[0], self.inp_res[1], self.replica)
            ent = self._plans[bucket] = (plan, self.net._plans)
        return ent[0]

    def _crop(self, frame_ptr, H, W, C, params_ptr, n, x_ptr, sh):
        """One crop launch: n boxes of one frame into the plan's input at x_ptr."""
        if self.cv2_exact:
            check(self.lib.ft_crop_affine_cv2_fwd(frame_ptr, H, W, C, params_ptr, n, self.inp_res[0], self.inp_res[1], self.mean_ptr,
                                                  self.inv_std_ptr, self.pre, None, x_ptr, sh), "ft_crop_affine_cv2_fwd")
        else:
            check(self.lib.ft_crop_affine_fwd(frame_ptr, H, W, C, params_ptr, n, self.inp_res[0], self.inp_res[1], self.mean_ptr,
                                              self.inv_std_ptr, self.pre, x_ptr, sh), "ft_crop_affine_fwd")

    def close(self) -> None:
        """Destroy the slots' events (ft_event_destroy) and drop the pinned buffers; the runner is unusable afterwards."""
        for sl in self.slots.values():
            if sl["pending"]:
                self.lib.ft_event_synchronize(sl["event"])
            self.lib.ft_event_destroy(sl["event"])
        self.slots = {}
        self._plans = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _launch(self, sl, plan, sh):
        """Plan replay + rows back into the pinned buffer + event, all through the C ABI on stream `sh`: a captured plan is one
        ft_graph_launch; the first call of a plan (eager run, tile benchmark, capture) goes through the model."""
        prog = plan.prog
        if plan.runs == 0 or prog.graph_exec is None or torch.cuda.current_device() != self.dev.index:
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    self.net.replay(plan)
            else:
                self.net.replay(plan)
        else:
            check(self.lib.ft_graph_launch(prog.graph_exec, sh), "ft_graph_launch")
            plan.runs += 1
        check(self.lib.ft_memcpy_async(sl["rows_ptr"], plan.kp_rows.data_ptr(), sl["rows_bytes"], sh), "ft_memcpy_async")
        check(self.lib.ft_event_record(sl["event"], sh), "ft_event_record")
        sl["pending"] = True

    def submit(self, frame_dev: torch.Tensor, boxes: np.ndarray):
        boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
        n = len(boxes)
        if n == 0:
            return None
        if frame_dev.dtype != torch.uint8 or frame_dev.dim() != 3 or not frame_dev.is_contiguous():
            raise ValueError("frame must be a contiguous uint8 [H,W,C] device tensor")
        bucket = next((b for b in self.BUCKETS if b >= n), None)
        if bucket is None:                                  # more boxes than the largest plan: chunks, each waited for in turn
            big = self.BUCKETS[-1]
            return ("chunks", [PoseRunner.result(self, PoseRunner.submit(self, frame_dev, boxes[lo:lo + big])) for lo in range(0, n, big)])
        centers, scales = boxes_to_center_scale(boxes, self.inp_res)
        sl = self._slot(bucket)
        self._fill_params(sl, centers, scales, n, bucket)
        H, W, C = frame_dev.shape
        plan = self._plan(bucket)
        sh = self._stream_handle()
        self._crop(frame_dev.data_ptr(), H, W, C, sl["params_ptr"], bucket, plan.x_static.data_ptr(), sh)
        self._launch(sl, plan, sh)
        return (sl, n, centers, scales, (plan.heatmaps.shape[2], plan.heatmaps.shape[3]))

    def submit_frames(self, frames_dev, boxes_list):
        """The boxes of SEVERAL frames in one network call (phase 2 of the clip pipeline: ~5 detector boxes per frame would
        otherwise pay one plan replay and one round trip per frame): one crop launch per frame, each writing its slice of the
        plan's input.  result() returns the rows of all boxes in order."""
        if self.stream is not None and torch.cuda.current_stream(self.dev) != self.stream:
            with torch.cuda.stream(self.stream):
                return PoseRunner.submit_frames(self, frames_dev, boxes_list)
        per = [np.asarray(b, dtype=np.float64).reshape(-1, 4) for b in boxes_list]
        total = sum(len(b) for b in per)
        if total == 0:
            return None
        bucket = next((b for b in self.BUCKETS if b >= total), None)
        if bucket is None:                                  # split the frame list in halves until each fits a plan
            if len(per) == 1:                               # (the base-class forms: GroupPoseRunner overrides submit / result)
                return PoseRunner.submit(self, frames_dev[0], per[0])
            h = len(per) // 2
            return ("chunks", [PoseRunner.result(self, PoseRunner.submit_frames(self, frames_dev[:h], per[:h])),
                               PoseRunner.result(self, PoseRunner.submit_frames(self, frames_dev[h:], per[h:]))])
        allb = np.concatenate(per)
        centers, scales = boxes_to_center_scale(allb, self.inp_res)
        sl = self._slot(bucket)
        self._fill_params(sl, centers, scales, total, total)
        plan = self._plan(bucket)
        sh = self._stream_handle()
        x = plan.x_static
        if bucket > total:
            x[total:].zero_()
        lo = 0
        for frame, b in zip(frames_dev, per):
            if len(b) == 0:
                continue
            H, W, C = frame.shape
            self._crop(frame.data_ptr(), H, W, C, sl["params_host"][lo:].data_ptr(), len(b), x[lo:].data_ptr(), sh)
            lo += len(b)
        self._launch(sl, plan, sh)
        return (sl, total, centers, scales, (plan.heatmaps.shape[2], plan.heatmaps.shape[3]))

    def result(self, handle) -> np.ndarray:
        if handle is None:
            return np.zeros((0, self.K, 3), dtype=np.float32)
        if handle[0] == "chunks":
            return np.concatenate(handle[1], axis=0)
        sl, n, centers, scales, hm_hw = handle
        check(self.lib.ft_event_synchronize(sl["event"]))
        sl["pending"] = False
        return heatmap_rows_to_image(sl["rows_np"][:n], centers, scales, hm_hw)

    def __call__(self, frame_dev: torch.Tensor, boxes: np.ndarray) -> np.ndarray:
        return self.result(self.submit(frame_dev, boxes))


class GroupPoseRunner(PoseRunner):
    """One PoseRunner shared by a GROUP of clips (tools/tracking/demo.run_clips): the members' per-frame submits of one round
    are collected and go through ONE plan replay when the scheduler calls flush() — the crops of, say, four clips' propagated
    boxes as one 16- or 32-crop batch instead of four 4- or 8-crop batches.  A small-batch replay is bound by its launch
    latencies (47 launches of 10-18 us at 8 crops) and replays on different streams barely overlap on the GPU, so batching
    across clips is what raises the GPU's throughput in the sequential passes; the flushed batch is exactly submit_frames()
    of the collected (frame, boxes) pairs, and a member's result() hands back its own rows.
    Protocol per round: member submits (any subset of the members, each at most once) -> flush() -> ... -> results.  A
    result() of a round that was not flushed yet flushes it (a member alone behaves like a plain runner)."""

    def __init__(self, net, inp_res=(256, 192), normalize=True, replica: int = 0, stream=None, cv2_exact: bool = False):
        super().__init__(net, inp_res, normalize, replica, stream, cv2_exact)
        self._pending = []                                 # [frame_dev, boxes [n,4]] of the round being collected
        self._round = {"handle": None, "cuts": None}       # the collecting round; replaced at flush

    def submit(self, frame_dev: torch.Tensor, boxes: np.ndarray):
        boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
        if len(boxes) == 0:
            return None
        self._pending.append((frame_dev, boxes))
        return ("member", self._round, len(self._pending) - 1)

    def flush(self) -> None:
        """Launch everything collected since the last flush as one batch (nothing collected: nothing happens)."""
        if not self._pending:
            return
        rnd, pend = self._round, self._pending
        self._pending, self._round = [], {"handle": None, "cuts": None}
        rnd["cuts"] = np.cumsum([0] + [len(b) for _, b in pend])
        rnd["handle"] = PoseRunner.submit_frames(self, [f for f, _ in pend], [b for _, b in pend])
        rnd["rows"] = None

    def result(self, handle) -> np.ndarray:
        if handle is None or handle[0] != "member":
            return PoseRunner.result(self, handle)
        _, rnd, i = handle
        if rnd["handle"] is None:
            if rnd is not self._round:
                raise RuntimeError("GroupPoseRunner: result() of a round that was dropped")
            self.flush()
        if rnd["rows"] is None:
            rnd["rows"] = PoseRunner.result(self, rnd["handle"])
        return rnd["rows"][rnd["cuts"][i]:rnd["cuts"][i + 1]]


def pose_est_frames(net, frames_dev, boxes_list, inp_res=(256, 192), normalize=True):
    """pose_est for the boxes of SEVERAL frames in one network call: crops of every frame (one ft_crop_affine_fwd launch
    per frame) are stacked into one batch, padded to the plan bucket, and leave through one final_preds.  Returns a list of
    [n_t,K,3] arrays.  (Phase 2 of the clip pipeline: ~5 boxes per frame would otherwise pay one plan replay, one arg-max
    and one device->host sync per frame.)"""
    counts = [len(np.asarray(b).reshape(-1, 4)) for b in boxes_list]
    total = sum(counts)
    if total == 0:
        return [np.zeros((0, _num_joints(net), 3), dtype=np.float32) for _ in counts]
    crops, cs, ss = [], [], []
    for frame, boxes in zip(frames_dev, boxes_list):
        boxes = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
        if len(boxes) == 0:
            continue
        c, s_ = boxes_to_center_scale(boxes, inp_res)
        crops.append(crop_boxes(frame, c, s_, inp_res, normalize))
        cs.append(c)
        ss.append(s_)
    batch = torch.cat(crops, 0)
    centers, scales = np.concatenate(cs), np.concatenate(ss)
    bucket = next(b for b in (8, 16, 32, 64, 128, 256, 1 << 30) if b >= total)
    if bucket > total and bucket < (1 << 30):
        batch = torch.cat((batch, batch.new_zeros((bucket - total,) + tuple(batch.shape[1:]))), 0)
    hm = net(batch)[:total]
    c, s_ = final_preds(hm, centers, scales, adjust_coords=True)
    kp = np.concatenate((c, s_), axis=2).astype(np.float32)
    out, lo = [], 0
    for n in counts:
        out.append(kp[lo:lo + n])
        lo += n
    return out


def pad_pairs_to_64(ims: torch.Tensor) -> torch.Tensor:
    """[B,3,2,H,W] -> [B,3,2,Hp,Wp] with Hp, Wp the next multiples of 64 (FlowNet's six stride-2 stages), filled by
    replicating the last row / column.  The reference feeds frames as they are (net_utils.py:73-92) and only works for
    sizes the net divides; zero padding would add a hard black edge and pull the net's own rgb_mean
    (lib/flownet/model/models.py:255) towards black, replication keeps both representative."""
    B, C, P, H, W = ims.shape
    Hp, Wp = -(-H // 64) * 64, -(-W // 64) * 64
    if (Hp, Wp) == (H, W):
        return ims
    return torch.nn.functional.pad(ims.reshape(B, C * P, H, W), (0, Wp - W, 0, Hp - H), mode="replicate").reshape(B, C, P, Hp, Wp)


def flow_est(net, prev_frame: torch.Tensor, cur_frame: torch.Tensor) -> np.ndarray:
    """prev / cur: uint8 [H,W,3] BGR (device or host) -> flow [2,H,W] fp32 numpy (net_utils.py:73-92):
    BGR -> RGB, pack [1,3,2,H,W] float 0..255, edge-replicate to a multiple of 64, crop the flow back."""
    dev = next(net.parameters()).device
    pair = torch.stack((torch.as_tensor(prev_frame).to(dev), torch.as_tensor(cur_frame).to(dev)))   # [2,H,W,3]
    H, W = pair.shape[1:3]
    ims = pad_pairs_to_64(pair.flip(-1).permute(3, 0, 1, 2).float().unsqueeze(0))
    return net(ims)[0, :, :H, :W].cpu().numpy()
