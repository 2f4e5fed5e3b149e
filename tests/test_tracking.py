"""Tracking glue (next rows N1-N3): host logic vs the oracle restatements on CPU; GPU crop kernel + pipeline on GPU."""
import numpy as np
import pytest
import torch

from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.tracking import FlowTracker, box_propagation, detect, nms
from flowtrack.pytorch_amd.tracking import net_utils, tracker
from oracle import tracking_ref


def _kp(seed, n=4, H=96, W=128):
    xy = synth.uniform01(seed, "kp", (n, 17, 2)) * [W - 10, H - 10] + 5
    s = (synth.uniform01(seed, "kps", (n, 17, 1)) > 0.25) * 0.8
    return np.concatenate((xy, s), axis=2)


def test_box_propagation_matches_restated_semantics():
    kp = _kp(1)
    kp[0, :, 2] = 0.7                                  # a fully visible person
    kp[1, 5:, 2] = 0.0                                 # mostly hidden
    flow = synth.normal(1, "flow", (2, 96, 128), std=4.0).numpy()
    got, want = box_propagation(kp, flow), tracking_ref.box_propagation_ref(kp, flow)
    assert got.shape == (4, 4) and np.allclose(got, want)
    # constant flow (dx, dy) = (3, -2): the box is the shifted keypoint hull grown by 15 %, clipped to the frame
    flow[:] = 0; flow[0] = 3.0; flow[1] = -2.0
    b = box_propagation(kp[:1], flow)[0]
    hull = np.array([kp[0, :, 0].min() + 3, kp[0, :, 1].min() - 2, kp[0, :, 0].max() + 3, kp[0, :, 1].max() - 2])
    ext = np.array([hull[2] - hull[0], hull[3] - hull[1]]) * 0.075
    want = np.concatenate((np.maximum(hull[:2] - ext, 0), np.minimum(hull[2:] + ext, [127, 95])))
    assert np.allclose(b, want)


def test_nms_matches_reference_loop_and_detect_unions():
    rng = np.random.RandomState(3)
    d = np.concatenate((rng.uniform(0, 50, (40, 2)), rng.uniform(50, 100, (40, 2)), rng.uniform(0, 1, (40, 1))), 1).astype(np.float32)
    for thr in (0.3, 0.5, 0.9):
        assert list(nms(d, thr)) == tracking_ref.nms_ref(d, thr)
    a = np.array([[10, 10, 50, 90, 0.9]], np.float32)
    dup = np.array([[12, 11, 52, 91, 0.6], [200, 40, 240, 120, 0.5]], np.float32)   # near-duplicate + a new box
    out = detect(a, 0.3, dup)
    assert out.shape == (2, 5) and out[0, 4] == np.float32(0.9) and out[1, 0] == 200
    # IoU exactly at the threshold suppresses (>=, nms.c:59): boxes [0,0,9,9] and [0,0,9,19] -> IoU = 100/200
    e = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 19, 0.8]], np.float32)
    assert list(nms(e, 0.5)) == [0] and list(nms(e, 0.51)) == [0, 1]
    assert nms(np.zeros((0, 5), np.float32), 0.3).size == 0


def test_tracker_keeps_ids_under_flow_and_spawns_new_ones():
    kp = _kp(5, n=3)
    kp[..., 2] = 0.9
    boxes = np.array([[k[:, 0].min(), k[:, 1].min(), k[:, 0].max(), k[:, 1].max(), 0.9] for k in kp])
    tr = FlowTracker(oks_threshold=0.5)
    assert tr.update(kp, boxes) == [0, 1, 2]
    flow = np.zeros((2, 96, 128), np.float32); flow[0] = 4.0; flow[1] = 1.0
    moved = kp.copy(); moved[..., 0] += 4.0; moved[..., 1] += 1.0
    perm = [2, 0, 1]
    ids = tr.update(moved[perm], boxes[perm] + [4, 1, 4, 1, 0], flow)           # same people, shuffled, displaced by the flow
    assert ids == [2, 0, 1]
    far = moved.copy(); far[0, :, 0] = 120 - far[0, :, 0]                        # person 0 replaced by someone else far away
    ids = tr.update(np.concatenate((far[:1], moved[1:])), boxes + [4, 1, 4, 1, 0], np.zeros_like(flow))
    assert ids[1:] == [1, 2] and ids[0] == 3
    assert np.isclose(tracker.pose_oks(kp[0], kp[0], 1000.0), 1.0)


def test_tracker_survives_a_frame_without_detections():
    """A frame where no box survives (pose_est returns (0,17,3)): no ids, tracks move with the flow and age; the people
    coming back one frame later (max_age=1) keep their ids, after two empty frames they are new."""
    kp = _kp(9, n=2)
    kp[..., 2] = 0.9
    boxes = np.array([[k[:, 0].min(), k[:, 1].min(), k[:, 0].max(), k[:, 1].max(), 0.9] for k in kp])
    flow = np.zeros((2, 96, 128), np.float32); flow[0] = 2.0
    tr = FlowTracker(oks_threshold=0.5, max_age=1)
    assert tr.update(kp, boxes) == [0, 1]
    assert tr.update(np.zeros((0, 17, 3)), np.zeros((0, 5)), flow) == []
    assert sorted(tr.tracks) == [0, 1] and all(t["age"] == 1 for t in tr.tracks.values())
    assert np.allclose(tr.tracks[0]["kpts"][:, 0], kp[0, :, 0] + 2.0)              # still propagated
    moved = kp.copy(); moved[..., 0] += 4.0
    assert tr.update(moved, boxes + [4, 0, 4, 0, 0], flow) == [0, 1]
    assert tr.update(np.zeros((0, 17, 3)), np.zeros((0,)), None) == []             # degenerate box array shapes too
    assert tr.update([], [], flow) == [] and tr.tracks == {}                        # aged out after two empty frames
    assert tr.update(moved, boxes, flow) == [2, 3]


def test_oks_matrix_equals_the_pairwise_loop_and_matching_is_unchanged():
    """pose_oks_matrix (one broadcast) == pose_oks per pair, incl. joints below the score threshold and empty sets; the
    greedy matcher on top of it assigns what the per-pair loop assigned (random poses, several frames)."""
    rng = np.random.default_rng(3)
    dets = rng.normal(50, 20, (7, 17, 3)); trks = rng.normal(50, 20, (5, 17, 3))
    dets[..., 2] = rng.uniform(-0.2, 1, (7, 17)); trks[..., 2] = rng.uniform(-0.2, 1, (5, 17))
    trks[2, :, 2] = -1.0                                                  # a track with no countable joint
    areas = rng.uniform(500, 5000, 7)
    got = tracker.pose_oks_matrix(dets, trks, areas, kpt_thresh=0.1)
    want = np.array([[tracker.pose_oks(d, t, a, kpt_thresh=0.1) for t in trks] for d, a in zip(dets, areas)])
    assert np.allclose(got, want, rtol=0, atol=1e-15) and np.all(got[:, 2] == 0.0)
    assert tracker.pose_oks_matrix(dets[:0], trks, areas[:0]).shape == (0, 5)

    def loop_update(tr, keypoints, boxes):                               # the matcher as first written: per-pair loop
        ids = list(tr.tracks)
        moved = np.stack([tr.tracks[i]["kpts"] for i in ids]) if ids else np.zeros((0,) + keypoints.shape[1:])
        order = np.argsort(-boxes[:, 4], kind="stable")
        taken, assigned = set(), [-1] * len(keypoints)
        nxt = tr.next_id
        for d in order:
            area = max((boxes[d, 2] - boxes[d, 0]) * (boxes[d, 3] - boxes[d, 1]), 1.0)
            best, best_s = None, tr.oks_threshold
            for ti, tid in enumerate(ids):
                if tid in taken:
                    continue
                s_ = tracker.pose_oks(keypoints[d], moved[ti], area, kpt_thresh=tr.kpt_threshold)
                if s_ > best_s:
                    best, best_s = tid, s_
            if best is None:
                best, nxt = nxt, nxt + 1
            taken.add(best)
            assigned[d] = best
        return assigned

    tr = FlowTracker(oks_threshold=0.3)
    base = rng.normal(60, 25, (6, 17, 3)); base[..., 2] = 0.8
    for t in range(6):
        kp = base + rng.normal(0, 1.5, base.shape) + t
        kp[..., 2] = 0.8
        kp = kp[rng.permutation(6)][: 6 - (t % 2)]
        boxes = np.array([[k[:, 0].min(), k[:, 1].min(), k[:, 0].max(), k[:, 1].max(), s_] for k, s_ in zip(kp, rng.uniform(0.3, 1, len(kp)))])
        want_ids = loop_update(tr, kp, boxes)
        assert tr.update(kp, boxes) == want_ids


def test_boxes_to_center_scale():
    c, s = net_utils.boxes_to_center_scale(np.array([[10, 20, 50, 180], [0, 0, 191, 100]], float), (256, 192))
    assert np.allclose(c, [[30, 100], [95.5, 50]]) and np.allclose(s, [160, 191 / 192 * 256])


@pytest.mark.gpu
def test_crop_kernel_matches_oracle(hip_lib):
    H, W = 120, 160
    img = (synth.uniform01(2, "img", (H, W, 3)) * 255).astype(np.uint8)
    centers = np.array([[80.0, 60.0], [5.0, 10.0], [150.5, 110.25]])
    scales = np.array([100.0, 64.0, 300.0])
    dev = torch.from_numpy(img).cuda()
    for normalize in (False, True):
        got = net_utils.crop_boxes(dev, centers, scales, (64, 48), normalize=normalize).cpu().numpy()
        for i in range(3):
            kw = dict(mean=net_utils.BGR_MEAN, inv_std=[1 / s for s in net_utils.BGR_STD], pre_scale=1 / 255.0) if normalize else {}
            want = tracking_ref.crop_affine_ref(img, centers[i], scales[i], (64, 48), **kw)
            assert np.abs(got[i] - want).max() <= (2e-3 if normalize else 0.05), (i, normalize)
    # identity geometry: scale == crop height and centre at the crop centre reproduce the pixels; outside the frame is 0
    got = net_utils.crop_boxes(dev, np.array([[24.0, 32.0]]), np.array([64.0]), (64, 48), normalize=False).cpu().numpy()[0]
    assert np.array_equal(got.transpose(1, 2, 0), img[:64, :48].astype(np.float32))
    got = net_utils.crop_boxes(dev, np.array([[-500.0, -500.0]]), np.array([64.0]), (64, 48), normalize=False)
    assert torch.all(got == 0)


@pytest.mark.gpu
def test_crop_cv2_exact_kernel_is_bit_exact(hip_lib):
    """ft_crop_affine_cv2_fwd == OpenCV's uint8 warpAffine as restated in oracle/tracking_ref.py::warp_affine_cv2_ref, BIT FOR
    BIT (integer work): boxes inside, across and beyond the frame's edges, up- and down-scaling, 1 / 3 / 4 channels, and
    general (rotated / sheared) matrices through the C ABI; the fp32 output is the normalised uint8 value."""
    import ctypes
    rng = np.random.default_rng(11)
    lib = hip_lib
    for (H, W, C) in ((120, 160, 3), (67, 45, 1), (90, 130, 4)):
        img = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
        dev = torch.from_numpy(img).cuda()
        centers = np.concatenate((rng.uniform(-20, [W + 20, H + 20], (12, 2)), [[W / 2, H / 2], [0.0, 0.0], [W - 1.0, H - 1.0], [-500.0, -500.0]]))
        scales = np.concatenate((rng.uniform(8, 400, 12), [64.0, 64.0, 1000.0, 64.0]))
        if C == 3:
            f32, u8 = net_utils.crop_boxes(dev, centers, scales, (64, 48), normalize=True, cv2_exact=True, return_u8=True)
            f32, u8 = f32.cpu().numpy(), u8.cpu().numpy()
        else:
            minv = torch.from_numpy(net_utils.cv2_crop_matrices(centers, scales, (64, 48))).cuda()
            u8t = torch.empty((len(scales), 64, 48, C), dtype=torch.uint8, device="cuda")
            assert lib.ft_crop_affine_cv2_fwd(dev.data_ptr(), H, W, C, minv.data_ptr(), len(scales), 64, 48, None, None, ctypes.c_float(1.0),
                                              u8t.data_ptr(), None, None) == 0
            torch.cuda.synchronize()
            u8 = u8t.cpu().numpy()
        for i in range(len(scales)):
            want = tracking_ref.crop_cv2_ref(img, centers[i], scales[i], (64, 48))               # [C, h, w] uint8
            assert np.array_equal(u8[i].transpose(2, 0, 1), want), (H, W, C, i)
            if C == 3:
                norm = (want.astype(np.float32) * np.float32(1 / 255.0) - np.asarray(net_utils.BGR_MEAN, np.float32)[:, None, None]) \
                    * (1 / np.asarray(net_utils.BGR_STD, np.float32))[:, None, None]
                assert np.abs(f32[i] - norm).max() <= 2e-6
        assert not u8[-1].any()
    # general 2x3 matrices (rotation + shear + zoom), handed over as cv2 would invert them
    img = rng.integers(0, 256, (80, 100, 3), dtype=np.uint8)
    dev = torch.from_numpy(img).cuda()
    mats = []
    for k in range(6):
        th = rng.uniform(-np.pi, np.pi)
        z = rng.uniform(0.3, 3.0)
        mats.append(tracking_ref.cv2_invert_affine([z * np.cos(th), -z * np.sin(th) + 0.1, rng.uniform(-30, 60), z * np.sin(th), z * np.cos(th), rng.uniform(-30, 60)]))
    minv = torch.from_numpy(np.stack(mats)).cuda()
    u8t = torch.empty((6, 50, 70, 3), dtype=torch.uint8, device="cuda")
    assert lib.ft_crop_affine_cv2_fwd(dev.data_ptr(), 80, 100, 3, minv.data_ptr(), 6, 50, 70, None, None, ctypes.c_float(1.0), u8t.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    for k in range(6):
        assert np.array_equal(u8t[k].cpu().numpy(), tracking_ref.warp_affine_cv2_ref(img, mats[k], (50, 70))), k
    assert lib.ft_crop_affine_cv2_fwd(dev.data_ptr(), 80, 100, 3, minv.data_ptr(), 6, 50, 70, None, None, ctypes.c_float(1.0), None, None, None) != 0


@pytest.mark.gpu
def test_pose_runner_cv2_exact_crops_feed_the_net_and_keypoints_agree(hip_lib):
    """PoseRunner(cv2_exact=True): the plan's input is the normalised cv2-exact uint8 crop (bit for bit the values crop_boxes
    gives), and — the condition for keeping the fp32 crop as the fast default — the key points of the two crop forms agree to a
    small fraction of a heat-map pixel on a smooth frame (the two crops differ by <= ~0.6 grey levels there)."""
    from flowtrack.pytorch_amd.pose import models as pose_models
    from flowtrack.pytorch_amd.tracking import PoseRunner
    dev = torch.device("cuda", 0)
    net = pose_models.deconv("resnet50", num_classes=17, pretrained=False)
    net.load_state_dict(synth.fill_pose_state_dict(net.state_dict(), 3))
    net = net.to(dev).eval()
    net.compute_dtype = torch.float32
    yy, xx = np.meshgrid(np.arange(384), np.arange(512), indexing="ij")
    frame_np = np.stack([127 + 100 * np.sin(xx / 37.0) * np.cos(yy / 29.0), (xx + yy) * 0.28, 255 - xx * 0.45], -1).clip(0, 255).astype(np.uint8)
    frame = torch.from_numpy(frame_np).to(dev)
    boxes = np.array([[30, 40, 130, 300], [200, 10, 330, 380], [400, 100, 500, 250], [5, 5, 60, 90], [250, 200, 300, 260]], dtype=np.float64)
    exact, fast = PoseRunner(net, cv2_exact=True), PoseRunner(net)
    h = exact.submit(frame, boxes)
    kp_exact = exact.result(h)
    centers, scales = net_utils.boxes_to_center_scale(boxes)
    want_x = net_utils.crop_boxes(frame, centers, scales, cv2_exact=True)
    plan = net.plan_for(8, 256, 192)
    assert torch.equal(plan.x_static[:5], want_x)
    u8 = net_utils.crop_boxes(frame, centers, scales, return_u8=True)[1].cpu().numpy()
    assert np.array_equal(u8[1].transpose(2, 0, 1), tracking_ref.crop_cv2_ref(frame_np, centers[1], scales[1], (256, 192)))
    kp_fast = fast(frame, boxes)
    hm_px = (scales / 64.0)[:, None]                                            # one heat-map pixel in image pixels, per box
    d = np.abs(kp_exact[..., :2] - kp_fast[..., :2]).max(-1) / hm_px
    assert np.median(d) <= 0.26 and np.isfinite(kp_exact).all(), (np.median(d), d.max())
    exact.close()
    fast.close()
    assert exact.slots == {} and fast.slots == {}


@pytest.mark.gpu
def test_group_runner_more_boxes_than_the_largest_bucket(hip_lib):
    """the > largest-bucket fallbacks of submit / submit_frames must use the BASE-class submit / result — in a
    GroupPoseRunner the overridden submit() only collects.  257 boxes of one frame (and 5 + 6 over two frames with tiny buckets)
    through a group runner == the plain runner's rows."""
    from flowtrack.pytorch_amd.pose import models as pose_models
    from flowtrack.pytorch_amd.tracking import GroupPoseRunner, PoseRunner
    dev = torch.device("cuda", 0)
    net = pose_models.deconv("resnet50", num_classes=17, pretrained=False)
    net.load_state_dict(synth.fill_pose_state_dict(net.state_dict(), 3))
    net = net.to(dev).eval()
    net.compute_dtype = torch.float16
    frame = torch.from_numpy((synth.uniform01(6, "frame", (384, 512, 3)) * 255).astype(np.uint8)).to(dev)
    base = np.array([[30, 40, 130, 300], [200, 10, 330, 380], [400, 100, 500, 250], [5, 5, 60, 90], [250, 200, 300, 260]], dtype=np.float64)
    grp, ref = GroupPoseRunner(net, replica=1, stream=torch.cuda.Stream(device=dev)), PoseRunner(net)
    grp.BUCKETS = ref.BUCKETS = (4, 8)
    many = np.concatenate([base + k for k in range(4)])[:17]                    # 17 > 8: chunks of 8, 8, 1
    h = grp.submit(frame, many)
    grp.flush()
    got = grp.result(h)
    want = ref(frame, many)
    assert got.shape == (17, 17, 3) and np.array_equal(got, want)
    h1, h2 = grp.submit(frame, base), grp.submit(frame, (base + 3.0)[:4])        # 9 boxes over two members: split by frame list
    r1, r2 = grp.result(h1), grp.result(h2)
    assert np.array_equal(r1, ref(frame, base)) and np.array_equal(r2, ref(frame, (base + 3.0)[:4]))


@pytest.mark.gpu
def test_clip_pipeline_runs_and_tracks(hip_lib):
    """End-to-end plumbing of the video pipeline (synthetic weights: geometry is checked, not pose quality)."""
    import types
    from tools.tracking import demo
    args = types.SimpleNamespace(pose_backbone=50, pose_model="", flow_net="FlowNet2S", flow_model="", fp16=True)
    pose_net, flow_net = demo.build_nets(args, torch.device("cuda:0"))
    frames, dets = demo.synthetic_clip(6, H=192, W=256, n_people=3, seed=4)
    out, tm = demo.run_clip(frames, dets, pose_net, flow_net)
    assert len(out) == 6 and all(f["keypoints"].shape[1:] == (17, 3) for f in out)
    assert all(len(f["ids"]) == len(f["boxes"]) >= 1 for f in out)
    assert all(np.isfinite(f["keypoints"]).all() for f in out)
    # flow_est wrapper: shape, finite, and == the batched path used by run_clip
    f = net_utils.flow_est(flow_net, frames[0], frames[1])
    assert f.shape == (2, 192, 256) and np.isfinite(f).all()


# ---- the clip pipeline (tools/tracking/demo.py) with stand-in networks: functional signal + sharding ----------------------
# heat-map peaks of the stand-in pose net, (column, row) in the 48 x 64 map: symmetric around the crop centre (24, 32) and
# spanning 56 rows / 28 columns, so that the key-point hull grown by 15 % (box_propagation) reproduces the crop box: the
# propagated box of a propagated box neither drifts nor shrinks
_PEAKS = np.stack((10 + np.round(((np.arange(17) * 7) % 17) * 28 / 16.0), 4 + np.round(np.arange(17) * 56 / 16.0)), 1)
_REL = np.stack((0.21 + 0.58 * ((np.arange(17) * 7) % 17) / 16.0, 0.065 + 0.87 * np.arange(17) / 16.0), 1)   # (x, y) in a box


def _gt_pose(gt_boxes, boxes):
    """A perfect "pose net": every query box gets the 17 key points of the ground-truth person it overlaps most."""
    boxes = np.asarray(boxes, np.float64).reshape(-1, 4)
    out = np.zeros((len(boxes), 17, 3), np.float32)
    for i, b in enumerate(boxes):
        ix = np.maximum(0, np.minimum(b[2], gt_boxes[:, 2]) - np.maximum(b[0], gt_boxes[:, 0]))
        iy = np.maximum(0, np.minimum(b[3], gt_boxes[:, 3]) - np.maximum(b[1], gt_boxes[:, 1]))
        g = gt_boxes[int(np.argmax(ix * iy))]
        out[i, :, :2] = g[:2] + _REL * (g[2:4] - g[:2])
        out[i, :, 2] = 0.9
    return out


def _separated_people(T, H=384, W=512):
    """Five people who never overlap (box NMS cannot merge two of them), constant velocities, detector jitter +-3 px.
    Returns detector boxes, ground-truth boxes and the true flow fields (each person's velocity inside its box + 12 px)."""
    x0 = np.array([20.0, 120.0, 225.0, 330.0, 430.0])
    y0 = np.array([30.0, 180.0, 60.0, 150.0, 40.0])
    w = np.array([55.0, 60.0, 50.0, 58.0, 52.0])
    h = np.array([130.0, 150.0, 120.0, 160.0, 140.0])
    vx = np.array([0.15, -0.1, 0.1, -0.15, 0.1])
    vy = np.array([0.4, -0.5, 0.6, 0.3, -0.2])
    dets, gts = [], []
    flows = np.zeros((T - 1, 2, H, W), np.float32)
    for t in range(T):
        g = np.stack((x0 + vx * t, y0 + vy * t, x0 + vx * t + w, y0 + vy * t + h), 1)
        jit = (synth.uniform01(3, "jit%d" % t, (5, 4)) - 0.5) * 6.0
        score = 0.5 + 0.5 * synth.uniform01(3, "score%d" % t, (5, 1))
        gts.append(g)
        dets.append(np.concatenate((g + jit, score), 1).astype(np.float32))
        if t < T - 1:
            for i in range(5):
                xa, ya, xb, yb = (int(max(0, g[i, 0] - 12)), int(max(0, g[i, 1] - 12)), int(min(W, g[i, 2] + 12)), int(min(H, g[i, 3] + 12)))
                flows[t, 0, ya:yb, xa:xb] = vx[i]
                flows[t, 1, ya:yb, xa:xb] = vy[i]
    return dets, gts, flows


def test_tracking_pass_keeps_five_ids_for_five_people():
    """process_frame's union + NMS (tools/tracking/demo.py:35-42) + the flow tracker on the synthetic clip with a perfect
    pose stand-in and zero flow: the box count stays at the number of people and so does the number of ids."""
    from tools.tracking import demo
    T = 80
    dets, gts, flows = _separated_people(T)
    kp_det = [_gt_pose(gts[t], dets[t][:, :4]) for t in range(T)]
    out = demo.tracking_pass(dets, kp_det, flows, lambda t, boxes: _gt_pose(gts[t], boxes))
    assert all(len(f["boxes"]) == 5 for f in out), sorted({len(f["boxes"]) for f in out})
    ids = [tuple(sorted(f["ids"])) for f in out]
    assert len({i for f in out for i in f["ids"]}) == 5 and all(i == ids[0] for i in ids)
    # an untrained pose net (key points anywhere) cannot blow the per-frame work up: the union is capped
    rng = np.random.default_rng(0)
    junk = lambda t, boxes: np.concatenate((rng.uniform(0, 380, (len(boxes), 17, 2)), np.full((len(boxes), 17, 1), 0.5)), 2)
    out = demo.tracking_pass(dets, [junk(t, dets[t]) for t in range(T)], flows, junk, max_boxes=12)
    assert max(len(f["boxes"]) for f in out) <= 12


def test_tracking_pass_keeps_every_nms_survivor_when_the_detector_misses():
    """The reference's process_frame keeps `dets[keep]` for every NMS survivor above thresh (tools/tracking/demo.py:38-41): in
    a frame where the detector returns nothing, the boxes propagated from the previous poses carry all five tracks (and their
    ids) through; with two detections out of five the three propagated boxes still stay.  No default cap on the union."""
    from tools.tracking import demo
    T = 30
    dets, gts, flows = _separated_people(T)
    dets[10] = np.zeros((0, 5), dtype=np.float32)
    dets[20] = dets[20][:2]
    kp_det = [_gt_pose(gts[t], dets[t][:, :4]) for t in range(T)]
    out = demo.tracking_pass(dets, kp_det, flows, lambda t, boxes: _gt_pose(gts[t], boxes))
    assert [len(f["boxes"]) for f in out] == [5] * T
    ids = [tuple(sorted(f["ids"])) for f in out]
    assert all(i == ids[0] for i in ids) and len(set(ids[0])) == 5
    # the opt-in bound of the synthetic-weights demo ("2x" the detector boxes, at least 4) does drop tracks there
    capped = demo.tracking_pass(dets, kp_det, flows, lambda t, boxes: _gt_pose(gts[t], boxes), max_boxes="2x")
    assert len(capped[10]["boxes"]) == 4 and len(capped[20]["boxes"]) == 4


@pytest.mark.gpu
def test_clip_pipeline_functional_signal_on_gpu(hip_lib):
    """The GPU pieces of the clip pipeline (device-resident clip, ft_crop_affine_fwd crops, heat-map arg-max + inverse
    affine, flow fields read back for the propagation) around a stand-in pose module whose heat maps peak at fixed
    crop-relative points and a stand-in flow net that returns the true motion: five people keep five ids, one box each."""
    from tools.tracking import demo

    class PeakPose(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.anchor = torch.nn.Parameter(torch.zeros(1))
            hm = torch.zeros((17, 64, 48))
            for k in range(17):
                hm[k, int(_PEAKS[k, 1]), int(_PEAKS[k, 0])] = 1.0
            self.register_buffer("hm", hm)

        def forward(self, crops):
            return self.hm[None].expand(crops.shape[0], -1, -1, -1).contiguous()

    T = 40
    frames, _ = demo.synthetic_clip(T)
    dets, _, flows = _separated_people(T)
    pose = PeakPose().cuda().eval()
    nxt = [0]

    def true_flow(ims):                       # run_clip walks the pairs in order: hand out the true fields one by one
        b = ims.shape[0]
        f = torch.from_numpy(flows[nxt[0]:nxt[0] + b]).to(ims.device)
        nxt[0] += b
        return f
    out, tm = demo.run_clip(frames, dets, pose, None, flow_fn=true_flow)
    assert all(len(f["boxes"]) == 5 for f in out), sorted({len(f["boxes"]) for f in out})
    assert len({i for f in out for i in f["ids"]}) == 5
    k0 = out[0]["keypoints"]
    b0 = out[0]["boxes"]
    assert np.all(k0[..., 2] > 0.5) and np.all(k0[..., 0] >= b0[:, None, 0] - 16) and np.all(k0[..., 0] <= b0[:, None, 2] + 16)
    assert np.all(k0[..., 1] >= b0[:, None, 1] - 16) and np.all(k0[..., 1] <= b0[:, None, 3] + 16)


def test_flow_frames_are_edge_replicated_to_multiples_of_64():
    """pad_pairs_to_64: the padded area repeats the last row / column (no black border, the net's rgb_mean stays the
    frame's), frames that already divide are returned untouched."""
    from flowtrack.pytorch_amd.tracking.net_utils import pad_pairs_to_64
    x = torch.arange(2 * 3 * 2 * 70 * 100, dtype=torch.float32).reshape(2, 3, 2, 70, 100)
    y = pad_pairs_to_64(x)
    assert tuple(y.shape) == (2, 3, 2, 128, 128)
    assert torch.equal(y[..., :70, :100], x)
    assert torch.equal(y[..., 70:, :100], x[..., 69:70, :].expand(-1, -1, -1, 58, -1))
    assert torch.equal(y[..., :70, 100:], x[..., :, 99:100].expand(-1, -1, -1, -1, 28))
    z = torch.zeros(1, 3, 2, 64, 128)
    assert pad_pairs_to_64(z) is z


@pytest.mark.gpu
def test_clip_300_frames_with_the_real_nets_is_deterministic_and_bounded(hip_lib):
    """BASELINE configs[4] at size on one GPU: the 300-frame 512x384 synthetic clip through the REAL nets (ResNet-50 pose fp16,
    FlowNet2S fp16, synthetic weights) — flow of the 299 pairs, pose of the detector boxes, the sequential pass with the
    asynchronous per-frame pose runner (tools/tracking/demo.py:35-42, lib/tracking/net_utils.py:36-92).  Properties that hold
    whatever the weights are: every output finite and inside the frame, the per-frame box count within the "2x" bound the
    synthetic-weights demo opts into, ids unique per frame, and two runs identical in every box, key point and id."""
    import types
    from tools.tracking import demo
    args = types.SimpleNamespace(pose_backbone=50, pose_model="", flow_net="FlowNet2S", flow_model="", fp16=True)
    dev = torch.device("cuda", 0)
    pose, flow = demo.build_nets(args, dev)
    frames, dets = demo.synthetic_clip(300)
    assert frames.shape == (300, 384, 512, 3)
    runs = []
    for _ in range(2):
        out, tm = demo.run_clip(frames, dets, pose, flow, max_boxes="2x")
        assert len(out) == 300 and set(tm) >= {"flow_s", "pose_s", "track_s"}
        runs.append(out)
    for f, d in zip(runs[0], dets):
        assert 1 <= len(f["boxes"]) <= max(2 * len(d), 4)
        assert np.isfinite(f["boxes"]).all() and np.isfinite(f["keypoints"]).all()
        assert f["keypoints"].shape == (len(f["boxes"]), 17, 3) and len(f["ids"]) == len(f["boxes"])
        assert len(set(f["ids"])) == len(f["ids"]), "an id assigned twice in one frame"
        assert (f["boxes"][:, :2] >= -6).all() and (f["boxes"][:, 2] <= 511 + 6).all() and (f["boxes"][:, 3] <= 383 + 6).all()   # (detector jitter: +-3 px)
    for a, b in zip(*runs):
        assert np.array_equal(a["boxes"], b["boxes"]) and np.array_equal(a["keypoints"], b["keypoints"]) and list(a["ids"]) == list(b["ids"])


@pytest.mark.gpu
def test_pose_runner_matches_pose_est(hip_lib):
    """The asynchronous runner (crop into the plan's input, key-point rows inside the graph, closed-form inverse affine) gives
    what pose_est gives through final_preds (heat maps -> ft_heatmap_max_preds -> one 3x3 inverse per box), to 1e-3 px."""
    from flowtrack.pytorch_amd.pose import models as pose_models
    from flowtrack.pytorch_amd.tracking import PoseRunner, pose_est
    dev = torch.device("cuda", 0)
    net = pose_models.deconv("resnet50", num_classes=17, pretrained=False)
    net.load_state_dict(synth.fill_pose_state_dict(net.state_dict(), 11))
    net = net.to(dev).eval()
    net.compute_dtype = torch.float16
    frame = torch.from_numpy((synth.uniform01(5, "frame", (384, 512, 3)) * 255).astype(np.uint8)).to(dev)
    boxes = np.array([[30, 40, 130, 300], [200, 10, 330, 380], [400, 100, 500, 250], [5, 5, 60, 90], [250, 200, 300, 260]], dtype=np.float64)
    want = pose_est(net, frame, boxes, max_batch=8)
    runner = PoseRunner(net)
    got = runner(frame, boxes)
    assert got.shape == want.shape == (5, 17, 3)
    assert np.abs(got[..., :2] - want[..., :2]).max() <= 1e-3 and np.array_equal(got[..., 2], want[..., 2])
    # three boxes run in the 4-crop plan (other tile picks than the 8-crop plan: compare with pose_est in that same bucket)
    h1 = runner.submit(frame, boxes[:3])
    got3, want3 = runner.result(h1), pose_est(net, frame, boxes[:3], max_batch=8)
    assert np.abs(got3[..., :2] - want3[..., :2]).max() <= 1e-3 and np.array_equal(got3[..., 2], want3[..., 2])
    both = runner.result(runner.submit_frames([frame, frame], [boxes[:2], boxes[2:]]))
    assert np.array_equal(both, got)


def test_interleaved_passes_equal_the_passes_run_alone():
    """run_clips' scheduler on the CPU: three clips' tracking_pass_steps() generators advanced round-robin (one frame of each
    in turn) give, per clip, exactly what tracking_pass() gives alone — the passes share nothing but the host thread.  The
    asynchronous runner is a stand-in whose result() checks that every submit is collected exactly once, in order."""
    from tools.tracking import demo

    class _Runner:
        def __init__(self, gt):
            self.gt, self.pending = gt, []
        def submit(self, t, boxes):
            self.pending.append(t)
            return (t, np.array(boxes, dtype=np.float64))
        def result(self, h):
            assert self.pending.pop(0) == h[0]
            return _gt_pose(self.gt[h[0]], h[1])

    clips = []
    for c in range(3):
        T = 12 + 3 * c                                       # ragged lengths: a finished clip drops out of the rotation
        gt, dets, flows = _separated_people(T)
        dets = [d.copy() for d in dets]
        for d in dets:
            d[:, :4] += c                                    # three different clips
        kp_det = [_gt_pose(gt[t], dets[t][:, :4]) for t in range(T)]
        clips.append((gt, dets, kp_det, flows))
    alone = [demo.tracking_pass(d, k, f, _Runner(g)) for g, d, k, f in clips]
    gens = [demo.tracking_pass_steps(d, k, f, _Runner(g)) for g, d, k, f in clips]
    got, live = [None] * 3, [0, 1, 2]
    while live:
        for i in list(live):
            try:
                next(gens[i])
            except StopIteration as done:
                got[i] = done.value
                live.remove(i)
    for a, b in zip(alone, got):
        assert len(a) == len(b)
        for fa, fb in zip(a, b):
            assert np.array_equal(fa["boxes"], fb["boxes"]) and np.array_equal(fa["keypoints"], fb["keypoints"]) and list(fa["ids"]) == list(fb["ids"])


class _StubPoseNet(torch.nn.Module):
    """Stands in for DeconvResnet behind PoseRunner (plan_for / replay / kp_rows / x_static): the key-point rows of a crop are
    a fixed function of ONE pixel of that crop, so they do not depend on the batch the crop ran in — which makes "K clips
    grouped and interleaved" comparable bit for bit with "each clip alone" (the real nets' plans differ in fp16 rounding
    between batch sizes)."""
    num_classes = 17

    def __init__(self):
        super().__init__()
        self.anchor = torch.nn.Parameter(torch.zeros(1))
        self._plans = {}
        self.keypoints_in_plan = True
        self.replays = 0
        self.crops = 0

    def forward_keypoint_rows(self, x):                    # (run_clip picks the PoseRunner path by this attribute)
        raise NotImplementedError

    def plan_for(self, B, H, W, replica=0):
        import types
        key = (B, H, W, replica)
        if key not in self._plans:
            dev = self.anchor.device
            self._plans[key] = types.SimpleNamespace(
                x_static=torch.zeros((B, 3, H, W), dtype=torch.float32, device=dev), kp_rows=torch.zeros((B, 17, 3), device=dev),
                heatmaps=torch.empty((1, 17, H // 4, W // 4)), prog=types.SimpleNamespace(graph_exec=None), runs=0)
        return self._plans[key]

    def replay(self, plan):
        k = torch.arange(17, device=plan.x_static.device, dtype=torch.float32)
        v = plan.x_static[:, 0, 100, 80][:, None] + 0.5 * plan.x_static[:, 2, 160, 120][:, None]        # [B,1]: this crop's own pixels
        plan.kp_rows[:, :, 0] = 24.0 + 14.0 * torch.sin(0.7 * k[None] + 3.0 * v)
        plan.kp_rows[:, :, 1] = 32.0 + 22.0 * torch.cos(0.9 * k[None] + 2.0 * v)
        plan.kp_rows[:, :, 2] = 0.9
        plan.runs += 1
        self.replays += 1
        self.crops += plan.x_static.shape[0]
        return plan


@pytest.mark.gpu
def test_run_clips_grouped_equals_each_clip_alone(hip_lib):
    """configs[4] as throughput: K clips on one GPU (tools/tracking/demo.run_clips: passes interleaved, each group's per-frame
    crops in one plan replay, the next clip's batched phases underneath) give, clip by clip, the boxes / key points / ids of
    run_clip on that clip alone (tools/tracking/demo.py:35-42 is one clip, one frame at a time).  The crop kernel, the pinned
    slots, the streams and the offsets into the grouped batch are the real ones; the pose plan is a stand-in whose rows depend
    on the crop only, so the comparison is exact.  Grouping shows in the replay count: fewer, larger batches."""
    from tools.tracking import demo
    dev = torch.device("cuda", 0)
    pose = _StubPoseNet().to(dev).eval()
    flow = lambda ims: torch.full((ims.shape[0], 2, ims.shape[3], ims.shape[4]), 0.75, device=ims.device)   # noqa: E731
    clips = [demo.synthetic_clip(20 + 5 * c, seed=c) for c in range(5)]
    alone = [demo.run_clip(f, d, pose, None, flow_fn=flow, max_boxes="2x")[0] for f, d in clips]
    replays_alone = pose.replays
    for mode, groups in ((True, 2), (True, 1), (True, 5), (False, 2)):
        pose.replays = 0
        outs, tm = demo.run_clips(clips, pose, flow, max_boxes="2x", interleave=mode, groups=groups)
        assert tm["pass_frames"] == sum(len(f) for f, _ in clips) and len(outs) == 5
        for a, b in zip(alone, outs):
            assert len(a) == len(b)
            for fa, fb in zip(a, b):
                assert np.array_equal(fa["boxes"], fb["boxes"]) and list(fa["ids"]) == list(fb["ids"])
                assert np.array_equal(fa["keypoints"], fb["keypoints"])
        assert pose.replays <= replays_alone, (pose.replays, replays_alone)     # grouping never adds replays


@pytest.mark.gpu
def test_group_runner_is_submit_frames_of_the_round_and_clips_are_deterministic(hip_lib):
    """GroupPoseRunner with the real ResNet-50 (fp16): the members' submits of a round, flushed, are bit for bit
    PoseRunner.submit_frames of the same (frame, boxes) list — same bucket, same plan arithmetic — and each member gets its own
    rows; a member alone (result without flush) behaves like a plain runner.  Then run_clips on the real nets: finite, inside
    the frame, ids unique per frame, and two runs identical."""
    import types
    from flowtrack.pytorch_amd.tracking import GroupPoseRunner, PoseRunner
    from tools.tracking import demo
    args = types.SimpleNamespace(pose_backbone=50, pose_model="", flow_net="FlowNet2S", flow_model="", fp16=True)
    dev = torch.device("cuda", 0)
    pose, flow = demo.build_nets(args, dev)
    frames = [torch.from_numpy((synth.uniform01(20 + i, "frame", (384, 512, 3)) * 255).astype(np.uint8)).to(dev) for i in range(3)]
    boxes = [np.array([[30, 40, 130, 300], [200, 10, 330, 380]], dtype=np.float64),
             np.array([[400, 100, 500, 250], [5, 5, 60, 90], [250, 200, 300, 260]], dtype=np.float64),
             np.array([[100, 50, 180, 330], [300, 120, 380, 360], [20, 200, 90, 370], [420, 20, 500, 200]], dtype=np.float64)]
    grp = GroupPoseRunner(pose, replica=1, stream=torch.cuda.Stream(device=dev))
    hs = [grp.submit(f, b) for f, b in zip(frames, boxes)]
    assert grp.submit(frames[0], np.zeros((0, 4))) is None
    grp.flush()
    got = [grp.result(h) for h in hs]
    ref = PoseRunner(pose)
    assert ref.result(None).shape == (0, 17, 3)
    flat = ref.result(ref.submit_frames(frames, boxes))
    cuts = np.cumsum([0] + [len(b) for b in boxes])
    for i, g in enumerate(got):
        assert g.shape == (len(boxes[i]), 17, 3) and np.array_equal(g, flat[cuts[i]:cuts[i + 1]])
    solo = grp.result(grp.submit(frames[1], boxes[1]))                          # no flush: result() flushes the round
    assert np.array_equal(solo, ref(frames[1], boxes[1]))
    clips = [demo.synthetic_clip(16 + 4 * c, seed=c) for c in range(4)]
    runs = [demo.run_clips(clips, pose, flow, max_boxes="2x")[0] for _ in range(2)]
    for out, (f, d) in zip(runs[0], clips):
        assert len(out) == len(f)
        for fr_, dt in zip(out, d):
            assert 1 <= len(fr_["boxes"]) <= max(2 * len(dt), 4) and np.isfinite(fr_["boxes"]).all() and np.isfinite(fr_["keypoints"]).all()
            assert len(set(fr_["ids"])) == len(fr_["ids"]) == len(fr_["boxes"])
    for ra, rb in zip(*runs):
        for a, b in zip(ra, rb):
            assert np.array_equal(a["boxes"], b["boxes"]) and np.array_equal(a["keypoints"], b["keypoints"]) and list(a["ids"]) == list(b["ids"])


@pytest.mark.gpu
def test_pose_runner_mpii_16_joints_slot_guard_and_chunks(hip_lib):
    """num_classes is the model API's (MPII = 16 joints, tools/pose/main.py:22,57): the runner's row buffers follow it.  A second
    submit on a bucket whose slot is still in flight waits for that launch instead of rewriting its pinned box parameters;
    more boxes than the largest bucket are chunked."""
    from flowtrack.pytorch_amd.pose import models as pose_models
    from flowtrack.pytorch_amd.tracking import PoseRunner, pose_est
    dev = torch.device("cuda", 0)
    net = pose_models.deconv("resnet50", num_classes=16, pretrained=False)
    net.load_state_dict(synth.fill_pose_state_dict(net.state_dict(), 12))
    net = net.to(dev).eval()
    net.compute_dtype = torch.float16
    frame = torch.from_numpy((synth.uniform01(6, "frame", (384, 512, 3)) * 255).astype(np.uint8)).to(dev)
    boxes = np.array([[30, 40, 130, 300], [200, 10, 330, 380], [400, 100, 500, 250], [5, 5, 60, 90], [250, 200, 300, 260]], dtype=np.float64)
    runner = PoseRunner(net)
    got = runner(frame, boxes)
    want = pose_est(net, frame, boxes, max_batch=8)
    assert got.shape == want.shape == (5, 16, 3)
    assert np.abs(got[..., :2] - want[..., :2]).max() <= 1e-3 and np.array_equal(got[..., 2], want[..., 2])
    assert runner.result(None).shape == (0, 16, 3)
    # slot guard: two submits of one bucket without a result() in between — the first launch's rows are what its boxes give
    h1 = runner.submit(frame, boxes)
    h2 = runner.submit(frame, boxes[::-1].copy())
    second = runner.result(h2)
    assert np.array_equal(second, got[::-1])
    assert h1[0]["pending"] is False
    # chunking above the largest bucket
    runner.BUCKETS = (4, 8)
    many = np.concatenate([boxes, boxes + 1.0, boxes + 2.0])[:11]
    chunked = runner(frame, many)
    assert chunked.shape == (11, 16, 3) and np.array_equal(chunked[:5], got)
