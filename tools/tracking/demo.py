"""FlowTrack video pipeline on the HIP path — the driver the reference left unfinished (tools/tracking/demo.py:35-42
defines process_frame but never calls it; README.md:14).  BASELINE.json configs[4]: detector boxes -> pose crops +
FlowNet box propagation on a synthetic clip.

    python tools/tracking/demo.py --frames 300 [--pose_backbone 50] [--flow_net FlowNet2S] [--fp16]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/tracking/demo.py --frames 300

Per frame (process_frame, demo.py:35-42): flow(prev, cur) -> propagate previous keypoints' boxes -> union with the
detector's boxes + box NMS -> pose for every kept box -> flow-based greedy id assignment.
Multi-GPU decomposition (one process per GPU): the two embarrassingly parallel parts are sharded —
  phase 1: optical flow of every (t-1, t) pair, pairs sharded by index, flows gathered to rank 0;
  phase 2: pose of the DETECTOR boxes of every frame, frames sharded by index, keypoints all-gathered;
  phase 3 (rank 0, sequential in t as the method requires): propagation, union + NMS, pose of the propagated-only
           boxes that survive NMS, tracking ids.
The person detector itself is out of scope: boxes come from the synthetic clip generator
(ground truth + jitter) or from --dets FILE.npz.  Checkpoints: --pose_model / --flow_model in the reference formats;
without them deterministic synthetic weights are used (outputs are then only structurally meaningful).
"""
from __future__ import absolute_import, division, print_function

import argparse
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from flowtrack.pytorch_amd import parallel, synth                                  # noqa: E402
from flowtrack.pytorch_amd.flownet import models as flow_models                     # noqa: E402
from flowtrack.pytorch_amd.pose import models as pose_models                        # noqa: E402
from flowtrack.pytorch_amd.tracking import FlowTracker, GroupPoseRunner, PoseRunner, box_propagation, detect, flow_est, net_utils, pose_est, pose_est_frames  # noqa: E402


def synthetic_clip(n_frames, H=384, W=512, n_people=5, seed=0):
    """Textured rectangles ("people") moving with constant velocity over a textured background.
    Returns frames uint8 [T,H,W,3] (BGR) and per-frame detector boxes [n,5] = ground truth + U(-3,3) px jitter."""
    u = lambda name, shape: synth.uniform01(seed, name, shape)
    bg = (u("bg", (H // 8 + 1, W // 8 + 1, 3)).repeat(8, 0).repeat(8, 1)[:H, :W] * 120 + 40)
    size = np.stack((40 + 50 * u("pw", (n_people,)), 90 + 90 * u("ph", (n_people,))), 1)          # w, h
    pos0 = np.stack((u("px", (n_people,)) * (W - 140) + 20, u("py", (n_people,)) * (H - 200) + 10), 1)
    vel = (u("vel", (n_people, 2)) - 0.5) * 6.0
    tex = [u("tex%d" % i, (int(size[i, 1]), int(size[i, 0]), 3)) * 200 + 30 for i in range(n_people)]
    frames = np.empty((n_frames, H, W, 3), dtype=np.uint8)
    dets = []
    for t in range(n_frames):
        img = bg.copy()
        boxes = []
        for i in range(n_people):
            x0, y0 = pos0[i] + vel[i] * t
            x0 = float(np.clip(x0, 0, W - size[i, 0] - 1))
            y0 = float(np.clip(y0, 0, H - size[i, 1] - 1))
            xi, yi = int(x0), int(y0)
            h, w = tex[i].shape[:2]
            img[yi:yi + h, xi:xi + w] = tex[i]
            jit = (synth.uniform01(seed, "jit%d_%d" % (t, i), (4,)) - 0.5) * 6.0
            score = 0.5 + 0.5 * float(synth.uniform01(seed, "score%d_%d" % (t, i), (1,))[0])
            boxes.append([x0 + jit[0], y0 + jit[1], x0 + w - 1 + jit[2], y0 + h - 1 + jit[3], score])
        frames[t] = img.astype(np.uint8)
        dets.append(np.asarray(boxes, dtype=np.float32))
    return frames, dets


def build_nets(args, device):
    pose = pose_models.deconv("resnet%d" % args.pose_backbone, num_classes=getattr(args, "pose_classes", 17), pretrained=False)
    if args.pose_model:
        pose.load_state_dict(torch.load(args.pose_model, map_location="cpu")["state_dict"])
    else:
        pose.load_state_dict(synth.fill_pose_state_dict(pose.state_dict(), 11))
    fargs = types.SimpleNamespace(rgb_max=255.0, fp16=args.fp16)
    flow = getattr(flow_models, args.flow_net)(fargs)
    if args.flow_model:
        flow.load_state_dict(torch.load(args.flow_model, map_location="cpu")["state_dict"])
    else:
        flow.load_state_dict(synth.fill_flow_state_dict(flow.state_dict(), 11))
    pose, flow = pose.to(device).eval(), flow.to(device).eval()
    if args.fp16:
        pose.compute_dtype = flow.compute_dtype = torch.float16
    return pose, flow


def _sync(dev):
    if torch.device(dev).type == "cuda":
        torch.cuda.synchronize(dev)


def tracking_pass_steps(dets, kp_det, flows, pose_boxes, thresh=0.3, max_boxes=None):
    """GENERATOR form of tracking_pass(): yields once per frame, between the submit of the frame's propagated boxes (GPU work
    enqueued) and the wait for their key points — the point where run_clips() switches to another clip, so that clip's host
    work and its small-batch GPU latency overlap this clip's replay.  The final value (StopIteration.value) is the frame list.
    The sequential part of the method (one frame after the other, rank 0): propagate the previous frame's poses by the
    flow, union with the detector boxes + box NMS (process_frame, tools/tracking/demo.py:35-42), pose of the propagated
    boxes that survive, flow-based greedy id assignment.
    dets[t]: [n,5] detector boxes; kp_det[t]: [n,K,3] their key points (K = the pose net's num_classes); flows: [T-1,2,H,W] (host array or anything
    indexable by t giving a [2,H,W] numpy field); pose_boxes: either a callable (t, boxes[m,4]) -> [m,17,3] for the
    propagated-only boxes, or an object with submit(t, boxes) -> handle and result(handle) -> [m,17,3] (the GPU runner:
    frame t's id assignment — host work that nothing downstream of the next propagation depends on — then runs while the GPU
    computes frame t + 1's poses).
    Every NMS survivor is kept, as in the reference (process_frame keeps `dets[keep]`, demo.py:40-41).  `max_boxes` is an
    explicit opt-in bound on the work of a frame (highest scores kept): an int, or "2x" = twice the frame's detector boxes
    (every detection plus one propagated box each).  Nothing in the reference's union ages a propagated box that keeps
    out-scoring the detector, so with an UNTRAINED pose net the box count can grow frame after frame; the synthetic-weights
    demo therefore passes "2x" (main(), --max_boxes), a trained model runs uncapped."""
    from flowtrack.pytorch_amd.tracking.flow_utils import nms
    tracker = FlowTracker()
    is_async = hasattr(pose_boxes, "submit")
    K = next((np.asarray(k).shape[1] for k in kp_det if len(k)), 17)     # key points per person: the pose net's num_classes
    out, prev_kp, prev_dets = [], None, None
    deferred = None                                        # (frame index, kps, boxes, flow) whose ids are still to be assigned
    def assign(item):
        out[item[0]]["ids"] = tracker.update(item[1], item[2], item[3])
    for t in range(len(dets)):
        cur = np.asarray(dets[t], dtype=np.float32).reshape(-1, 5)
        n_det = len(cur)
        src = np.arange(n_det)
        flow = None
        if prev_kp is not None and len(prev_kp):
            flow = np.asarray(flows[t - 1])
            prop = box_propagation(prev_kp, flow)                                         # flow_utils.py:7-35
            prop_dets = np.concatenate((prop, prev_dets[:, 4:5]), axis=1).astype(np.float32)  # demo.py:38
            allb = np.concatenate((cur, prop_dets), 0)
            keep = nms(allb, thresh)                                                      # (score-descending)
            if max_boxes is not None:
                keep = keep[:max(2 * n_det, 4) if max_boxes == "2x" else int(max_boxes)]
            cur, src = allb[keep], keep
        kps = np.zeros((len(cur), K, 3), dtype=np.float32)
        from_det = src < n_det
        kps[from_det] = np.asarray(kp_det[t])[src[from_det]]
        handle = None
        if (~from_det).any():                                                            # propagated-only boxes
            if is_async:
                handle = pose_boxes.submit(t, cur[~from_det, :4])
            else:
                kps[~from_det] = pose_boxes(t, cur[~from_det, :4])
        if deferred is not None:                           # the previous frame's ids: while the GPU works on this frame
            assign(deferred)
        yield t                                            # a scheduler may run other clips' frames here
        if handle is not None:
            kps[~from_det] = pose_boxes.result(handle)
        out.append({"boxes": cur, "keypoints": kps, "ids": None})
        deferred = (t, kps, cur, flow)
        prev_kp, prev_dets = kps, cur
    if deferred is not None:
        assign(deferred)
    return out


def tracking_pass(dets, kp_det, flows, pose_boxes, thresh=0.3, max_boxes=None):
    """The sequential pass of ONE clip (see tracking_pass_steps for the arguments): drives the generator to its end."""
    steps = tracking_pass_steps(dets, kp_det, flows, pose_boxes, thresh, max_boxes)
    while True:
        try:
            next(steps)
        except StopIteration as done:
            return done.value


def _batched_phases(frames, dets, pose_net, flow_net, rank, world, flow_batch, pose_frames, pose_fn, flow_fn, dev, runner):
    """Phases 1 and 2 of a clip (both batch-parallel, sharded over ranks): the flow of every (t-1, t) pair and the pose of
    every detector box.  Returns (frames on the device, flows as a pinned host tensor on rank 0 | None, key points [T,nmax,K,3]
    numpy, timing dict)."""
    T = len(frames)
    batched_pose = pose_fn is None
    fr = torch.from_numpy(frames).to(dev)                                     # clip resident in HBM
    tm = {}
    # ---- phase 1: flows of pairs (t-1, t), sharded by pair index ---------------------------------------
    t0 = time.perf_counter()
    lo, hi = parallel.shard_range(T - 1, rank, world)
    H, W = frames.shape[1:3]
    local = torch.empty((hi - lo, 2, H, W), dtype=torch.float32, device=dev)
    for b0 in range(lo, hi, flow_batch):
        b1 = min(hi, b0 + flow_batch)
        ims = torch.stack((fr[b0:b1].flip(-1).permute(0, 3, 1, 2).float(),                # BGR -> RGB (net_utils.py:83-87)
                           fr[b0 + 1:b1 + 1].flip(-1).permute(0, 3, 1, 2).float()), dim=2)
        local[b0 - lo:b1 - lo] = flow_fn(net_utils.pad_pairs_to_64(ims))[:, :, :H, :W]    # edge-replicated to multiples of 64
    flows = parallel.all_gather_rows(local, T - 1)
    # the tracker reads the fields on the host: pinned buffer, copy overlapped with phase 2
    flows_host = torch.empty(flows.shape, dtype=flows.dtype, pin_memory=dev.type == "cuda") if rank == 0 else None
    if rank == 0:
        flows_host.copy_(flows, non_blocking=True)
    _sync(dev)
    tm["flow_s"] = time.perf_counter() - t0
    # ---- phase 2: pose of the detector boxes, frames sharded by index -----------------------------------
    t0 = time.perf_counter()
    lo, hi = parallel.shard_range(T, rank, world)
    nmax = max(len(d) for d in dets)
    K = int(getattr(pose_net, "num_classes", 17))          # 17 COCO / 16 MPII (tools/pose/main.py:22,57)
    kp_local = torch.zeros((hi - lo, nmax, K, 3), dtype=torch.float32, device=dev)
    if batched_pose:       # the HIP networks: several frames' crops per network call
        for t0_ in range(lo, hi, pose_frames):
            ts = range(t0_, min(hi, t0_ + pose_frames))
            if runner is None:
                kps_ts = pose_est_frames(pose_net, [fr[t] for t in ts], [dets[t][:, :4] for t in ts])
            else:
                flat = runner.result(runner.submit_frames([fr[t] for t in ts], [dets[t][:, :4] for t in ts]))
                cuts = np.cumsum([0] + [len(dets[t]) for t in ts])
                kps_ts = [flat[cuts[i]:cuts[i + 1]] for i in range(len(ts))]
            for t, kp in zip(ts, kps_ts):
                kp_local[t - lo, :len(kp)] = torch.from_numpy(kp).to(dev)
    else:
        for t in range(lo, hi):
            kp = pose_fn(fr[t], dets[t][:, :4])
            kp_local[t - lo, :len(kp)] = torch.from_numpy(np.asarray(kp, dtype=np.float32)).to(dev)
    kp_all = parallel.all_gather_rows(kp_local, T).cpu().numpy()
    _sync(dev)
    tm["pose_s"] = time.perf_counter() - t0
    return fr, flows_host, kp_all, tm


def _frame_runner(runner, fr):
    """The PoseRunner addressed by frame index (what tracking_pass_steps calls)."""
    class _Frames:
        submit = staticmethod(lambda t, boxes: runner.submit(fr[t], boxes))
        result = staticmethod(runner.result)
    return _Frames


def run_clip(frames, dets, pose_net, flow_net, rank=0, world=1, thresh=0.3, flow_batch=16, pose_frames=6, pose_fn=None,
             flow_fn=None, device=None, max_boxes=None):
    """Returns (per-frame dict list on rank 0 | None elsewhere, timing dict).
    pose_fn(frame [H,W,3] uint8 tensor, boxes [n,4]) -> [n,K,3] and flow_fn(ims [b,3,2,Hp,Wp]) -> [b,2,Hp,Wp] default to
    the HIP networks (pose_est / flow_net); the CPU tests inject stand-ins to check the sharding (device="cpu")."""
    T = len(frames)
    dev = torch.device(device) if device is not None else next(pose_net.parameters()).device
    # the HIP pose net: one device round trip per call, asynchronous (PoseRunner); any other module that maps crops to heat maps
    # (the tests' stand-ins): the generic pose_est / pose_est_frames path
    runner = PoseRunner(pose_net) if (pose_fn is None and hasattr(pose_net, "forward_keypoint_rows")) else None
    batched = pose_fn is None
    if pose_fn is None and runner is None:
        pose_fn = lambda frame, boxes: pose_est(pose_net, frame, boxes, max_batch=8)     # noqa: E731
    if flow_fn is None:
        flow_fn = flow_net
    fr, flows_host, kp_all, tm = _batched_phases(frames, dets, pose_net, flow_net, rank, world, flow_batch, pose_frames,
                                                 None if batched else pose_fn, flow_fn, dev, runner)
    if rank != 0:
        return None, tm
    # ---- phase 3: sequential tracking pass ----------------------------------------------------------------
    t0 = time.perf_counter()
    flows_np = flows_host.numpy()
    if runner is not None:
        pose_boxes = _frame_runner(runner, fr)
    else:
        pose_boxes = lambda t, boxes: pose_fn(fr[t], boxes)  # noqa: E731
    out = tracking_pass(dets, [kp_all[t, :len(dets[t])] for t in range(T)], flows_np, pose_boxes, thresh, max_boxes)
    tm["track_s"] = time.perf_counter() - t0
    if runner is not None:
        runner.close()
    return out, tm


def _clip_pipeline(frames, dets, pose_net, flow_net, batch_runner, pass_runner, flow_batch, pose_frames, thresh, max_boxes):
    """One clip's whole pipeline as a GENERATOR for run_clips(): yields "batched" after every enqueued chunk of the two
    batch-parallel phases (flow of 16 pairs / pose of 6 frames' detector boxes: GPU-heavy, little host work) and "pass" once per
    frame of the sequential pass (host-heavy, small GPU launches).  Nothing in it waits for the device except where a result
    is needed: the flows travel to a pinned host buffer behind an event, the key points come back through the runners' pinned
    slots.  Returns the frame list (StopIteration.value)."""
    dev = next(pose_net.parameters()).device
    T = len(frames)
    H, W = frames.shape[1:3]
    fr = torch.from_numpy(frames).to(dev)                                      # clip resident in HBM
    yield "batched"
    flows_host = torch.empty((T - 1, 2, H, W), dtype=torch.float32, pin_memory=True)
    for b0 in range(0, T - 1, flow_batch):
        b1 = min(T - 1, b0 + flow_batch)
        ims = torch.stack((fr[b0:b1].flip(-1).permute(0, 3, 1, 2).float(),                # BGR -> RGB (net_utils.py:83-87)
                           fr[b0 + 1:b1 + 1].flip(-1).permute(0, 3, 1, 2).float()), dim=2)
        flows_host[b0:b1].copy_(flow_net(net_utils.pad_pairs_to_64(ims))[:, :, :H, :W], non_blocking=True)
        yield "batched"
    flows_ready = torch.cuda.Event()
    flows_ready.record()
    kp_det = [None] * T
    for t0_ in range(0, T, pose_frames):
        ts = range(t0_, min(T, t0_ + pose_frames))
        h = batch_runner.submit_frames([fr[t] for t in ts], [dets[t][:, :4] for t in ts])
        yield "batched"
        flat = batch_runner.result(h)
        cuts = np.cumsum([0] + [len(dets[t]) for t in ts])
        for i, t in enumerate(ts):
            kp_det[t] = flat[cuts[i]:cuts[i + 1]]
    flows_ready.synchronize()
    if pass_runner.stream is not None:
        pass_runner.stream.wait_stream(torch.cuda.current_stream(dev))        # the clip's frames were uploaded on this stream
    steps = tracking_pass_steps(dets, kp_det, flows_host.numpy(), _frame_runner(pass_runner, fr), thresh, max_boxes)
    while True:
        try:
            next(steps)
        except StopIteration as done:
            return done.value
        yield "pass"


def run_clips(clips, pose_net, flow_net, thresh=0.3, flow_batch=16, pose_frames=6, max_boxes=None, interleave=True, groups=2,
              waves=1):
    """K INDEPENDENT clips on one GPU as a throughput workload (BASELINE configs[4] scaled out by clip: one process per GPU
    x K clips, no exchange between clips or ranks).  clips: [(frames uint8 [T,H,W,3], dets list), ...].
    One clip's wall time is 3/4 sequential pass (frame t's propagated boxes need frame t-1's key points), and that pass is
    bound by the LATENCY of a small-batch pose replay plus the host's matching, not by throughput.  Here every clip is a
    generator (_clip_pipeline) and one host thread advances them round-robin:
      * the clips are split into `groups` groups; the sequential passes of a group's clips advance in lock-step, one frame of
        each per round, and their propagated boxes of that round go through ONE plan replay (GroupPoseRunner: own plan
        replicas, pinned slots and stream per group) — a 16- or 32-crop batch instead of four 4- or 8-crop ones (a small-batch
        replay is 47 launches of 10-18 us whatever it holds, and replays on different streams barely overlap on the GPU);
      * while one group's batch runs, the host does the other group's NMS / id assignment / submits: tracking_pass_steps()
        hands control back between a frame's submit and its result;
      * the batch-parallel phases (flow of all pairs, pose of the detector boxes: the shared flow plan and batch runner,
        GPU-bound) run one clip after the other ahead of the passes, so that every group is full from its first round; with
        `waves` > 1 the clips go in waves and the next wave's batched phases advance one chunk per round under the current
        wave's host-bound passes.
    The reference's loop is one clip, one frame at a time (tools/tracking/demo.py:35-42, lib/tracking/net_utils.py:36-92).
    interleave=False runs the same clips one after the other through the same code (the A/B baseline).  A clip's result is
    what run_clip gives for it up to the arithmetic of the plan its crops ran in (a 16-crop plan may use other tile variants
    than an 8-crop one: fp16 rounding differences, as between any two batch sizes).
    Returns (list of per-frame dict lists, one per clip; timing dict)."""
    dev = next(pose_net.parameters()).device
    K = len(clips)
    shared = PoseRunner(pose_net)
    t0 = time.perf_counter()
    G = max(1, min(groups, K)) if (interleave and K > 1) else 0
    group_runners = [GroupPoseRunner(pose_net, replica=g + 1, stream=torch.cuda.Stream(device=dev)) for g in range(G)]
    runners = [group_runners[i % G] for i in range(K)] if G else [shared] * K
    gens = [_clip_pipeline(f, d, pose_net, flow_net, shared, runners[i], flow_batch, pose_frames, thresh, max_boxes)
            for i, (f, d) in enumerate(clips)]
    results = [None] * K
    tm = {"pass_frames": 0, "batched_chunks": 0}
    if not G:
        for i, g in enumerate(gens):
            while True:
                try:
                    tm["pass_frames" if next(g) == "pass" else "batched_chunks"] += 1
                except StopIteration as done:
                    results[i] = done.value
                    break
    else:
        # wave by wave: the batch-parallel phases of a wave's clips run one clip after the other (they share the flow plan and the
        # batch runner, and are GPU-bound: nothing to interleave), then the wave's sequential passes advance in lock-step — all
        # of a group's clips submit, the group flushes ONE replay, the next group's host work overlaps it.  The NEXT wave's batched
        # phases advance by one chunk per round underneath (the host is the busy side of a pass round).
        W = max(1, min(waves, K))
        wave_of = [i * W // K for i in range(K)]
        staged = {}                                          # clip -> True once its generator has yielded its first "pass"

        def advance_batched(i):
            """One chunk of clip i's batched phases; True when the clip has reached its pass (or ended)."""
            try:
                stage = next(gens[i])
            except StopIteration as done:
                results[i] = done.value
                return True
            tm["pass_frames" if stage == "pass" else "batched_chunks"] += 1
            return stage == "pass"

        for i in [i for i in range(K) if wave_of[i] == 0]:
            while not advance_batched(i):
                pass
            staged[i] = True
        for w in range(W):
            active = [i for i in range(K) if wave_of[i] == w and results[i] is None]
            nxt = [i for i in range(K) if wave_of[i] == w + 1]
            while active:
                for g in range(G):
                    for i in [i for i in active if i % G == g]:
                        try:
                            next(gens[i])
                        except StopIteration as done:
                            results[i] = done.value
                            active.remove(i)
                            continue
                        tm["pass_frames"] += 1
                    group_runners[g].flush()                 # this group's submits of the round: one plan replay
                if nxt and advance_batched(nxt[0]):          # the next wave's batched phases, one chunk per round
                    staged[nxt.pop(0)] = True
            for i in nxt:                                    # whatever of the next wave is not staged yet
                while not advance_batched(i):
                    pass
                staged[i] = True
            for g in range(G):
                group_runners[g].flush()                     # (first-frame submits of the clips staged above)
    _sync(dev)
    tm["wall_s"] = time.perf_counter() - t0
    for r in [shared] + group_runners:                       # slot events / pinned buffers (they leaked per call)
        r.close()
    return results, tm


def main(argv=None):
    ap = argparse.ArgumentParser(description="Pose estimation + flow-based tracking in video (HIP path)")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--people", type=int, default=5)
    ap.add_argument("--pose_backbone", default=50, type=int, help="50, 101, 152")
    ap.add_argument("--pose_model", type=str, default="", help="pose checkpoint (ckpt['state_dict'])")
    ap.add_argument("--flow_net", type=str, default="FlowNet2S", help="FlowNet2S, FlowNet2C, FlowNet2CS")
    ap.add_argument("--flow_model", type=str, default="", help="optical flow checkpoint (ckpt['state_dict'])")
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--save", type=str, default="")
    ap.add_argument("--clips", type=int, default=1, help="K independent clips interleaved on this GPU (throughput mode, run_clips)")
    ap.add_argument("--groups", type=int, default=2, help="clip groups of run_clips (each group's per-frame crops share one plan replay)")
    ap.add_argument("--waves", type=int, default=1, help="waves of run_clips (the next wave's batched phases run under the current wave's passes)")
    ap.add_argument("--pose_classes", type=int, default=17, help="key points per person: 17 (COCO) or 16 (MPII)")
    ap.add_argument("--max_boxes", type=str, default="auto",
                    help="boxes kept per frame after NMS: 'none' (the reference: every survivor), an integer, '2x' = twice the "
                         "detector boxes; 'auto' = 'none' with --pose_model, '2x' with the synthetic (untrained) weights")
    args = ap.parse_args(argv)
    max_boxes = {"auto": None if args.pose_model else "2x", "none": None, "2x": "2x"}.get(args.max_boxes)
    if args.max_boxes not in ("auto", "none", "2x"):
        max_boxes = int(args.max_boxes)
    rank, local_rank, world = parallel.init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    pose_net, flow_net = build_nets(args, device)
    if args.clips > 1:
        if world != 1:
            raise SystemExit("--clips is the per-GPU throughput mode: run one process per GPU, each with its own clips")
        clips = [synthetic_clip(args.frames, n_people=args.people, seed=c) for c in range(args.clips)]
        run_clips(clips, pose_net, flow_net, max_boxes=max_boxes, groups=args.groups, waves=args.waves)  # warm-up: plans / graphs of every replica
        for mode in (False, True):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            outs, tm = run_clips(clips, pose_net, flow_net, max_boxes=max_boxes, interleave=mode, groups=args.groups, waves=args.waves)
            dt = time.perf_counter() - t0
            print("clips: {} x {} frames, {}: {:.3f} s = {:.1f} frames/s total".format(
                args.clips, args.frames, "interleaved" if mode else "one after the other", dt, args.clips * args.frames / dt))
        return 0
    frames, dets = synthetic_clip(args.frames, n_people=args.people)
    run_clip(frames, dets, pose_net, flow_net, rank, world, max_boxes=max_boxes)          # warm-up: every plan / graph the timed run replays
    parallel.barrier()
    t0 = time.perf_counter()
    out, tm = run_clip(frames, dets, pose_net, flow_net, rank, world, max_boxes=max_boxes)
    parallel.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        n_ids = len({i for f in out for i in f["ids"]})
        print("clip: {} frames {}x{} on {} GPU(s): {:.2f} s = {:.1f} frames/s (flow {:.2f} s, detector-box pose {:.2f} s, "
              "tracking pass {:.2f} s); {} track ids for {} people".format(
                  args.frames, frames.shape[2], frames.shape[1], world, dt, args.frames / dt, tm["flow_s"], tm["pose_s"],
                  tm["track_s"], n_ids, args.people))
        if args.save:
            np.savez_compressed(args.save, boxes=np.array([f["boxes"] for f in out], dtype=object),
                                ids=np.array([f["ids"] for f in out], dtype=object), allow_pickle=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
