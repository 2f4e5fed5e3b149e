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
