"""Whole pose network on the GPU (through the C ABI) vs the committed goldens written by the imported
reference (tests/golden/make_golden.py) and vs the CPU oracle run live on the same seeded inputs."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.pose import evaluation, models
from oracle import keypoints_ref, pose_ref

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "pose_golden.npz"))
SEED = int(G["seed"])


def _model(depth, dtype):
    m = models.deconv(f"resnet{depth}", num_classes=17, pretrained=False)
    sd = synth.fill_pose_state_dict(m.state_dict(), SEED)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = dtype
    return m, sd


@pytest.mark.parametrize("depth", [50, 101])
def test_pose_fp32_matches_reference_golden(hip_lib, depth):
    """north_star bar: <= 1e-3 max-abs on fp32 heatmaps and identical arg-max indices."""
    B, H, W = (int(v) for v in G[f"r{depth}_shape"])
    m, sd = _model(depth, torch.float32)
    x = synth.pose_crops(SEED, B, H, W)
    hm = m(x.cuda()).cpu()
    want = pose_ref.pose_forward(sd, x, depth=depth)          # live oracle (pinned == reference)
    err = (hm - want).abs().max().item()
    assert err <= 1e-3, f"r{depth} fp32 heatmap max abs err {err:.3e}"
    if depth == 50:
        gerr = np.abs(hm[:2].numpy() - G["r50_heatmaps_b2"]).max()
        assert gerr <= 1e-3, f"vs committed reference heatmaps: {gerr:.3e}"
    # arg-max bit-exact (integer result), scores within tolerance
    _, scores, coords = evaluation.heatmap_max_preds(hm.cuda(), adjust_coords=False)
    idx = (coords[..., 1] * hm.shape[3] + coords[..., 0]).long().cpu().numpy()
    pos = G[f"r{depth}_scores"][..., 0] > 0
    assert np.array_equal(idx[pos], G[f"r{depth}_idx"][pos]), "arg-max keypoint indices differ from the reference"
    assert np.abs(scores.cpu().numpy() - G[f"r{depth}_scores"]).max() <= 1e-3
    # second call replays the captured HIP graph: identical bits
    hm2 = m(x.cuda()).cpu()
    assert torch.equal(hm, hm2)


@pytest.mark.parametrize("adjust", [False, True])
def test_final_preds_device_path(hip_lib, adjust):
    """ft_heatmap_max_preds + host transform == reference final_preds (0.4 semantics) on reference heatmaps."""
    hm = torch.from_numpy(G["r50_heatmaps_b2"])
    center, scale = G["r50_center"][:2], G["r50_scale"][:2]
    coords, scores = evaluation.final_preds(hm.cuda(), center, scale, adjust_coords=adjust)
    assert np.allclose(coords, G[f"r50_final_coords_adjust{int(adjust)}"][:2], atol=1e-3)
    assert np.array_equal(scores, G["r50_scores"][:2])
    c2, s2 = evaluation.max_preds(hm.cuda())
    oc, os_, oi = keypoints_ref.max_preds_ref(hm.numpy())
    assert np.array_equal(c2, oc) and np.array_equal(s2, os_)


def test_max_preds_edge_cases(hip_lib):
    """ties -> first occurrence; all-negative map -> coords zeroed; border maxima get no nudge."""
    hm = torch.full((1, 4, 8, 6), -1.0)
    hm[0, 0, 2, 3] = 5.0
    hm[0, 0, 5, 1] = 5.0            # tie: the row-major first one (2,3) wins
    hm[0, 2, 0, 0] = 3.0            # border maximum
    hm[0, 3, 4, 2] = 2.0; hm[0, 3, 4, 3] = 1.0; hm[0, 3, 3, 2] = 1.5   # interior: nudge +x, -y
    idx, score, coords = evaluation.heatmap_max_preds(hm.cuda(), adjust_coords=True)
    idx, score, coords = idx.cpu().numpy()[0], score.cpu().numpy()[0, :, 0], coords.cpu().numpy()[0]
    assert idx[0] == 2 * 6 + 3 and score[0] == 5.0
    assert score[1] == -1.0 and tuple(coords[1]) == (0.0, 0.0) and idx[1] == 0
    assert tuple(coords[2]) == (0.0, 0.0) and idx[2] == 0 and score[2] == 3.0
    assert tuple(coords[3]) == (2.25, 3.75)
    oc, os_, oi, pre = keypoints_ref.final_preds_ref(hm.numpy(), np.array([[0.0, 0.0]]), np.array([8.0]), adjust_coords=True)
    assert np.array_equal(pre[0], coords)


def test_pose_fp16_vs_fp32_oracle(hip_lib):
    """fp16 storage / fp32 accumulate (configs C2): report error and arg-max agreement vs the fp32 CPU
    oracle.  Guards sit at ~3x what is measured (0.08 % of the heat-map range, 98.4-99.3 % identical arg-max — DESIGN §4,
    bench.py `parity`): max-abs <= 0.3 % of the range, >= 97 % identical arg-max, and every mismatch must be a near-tie
    (reference top-1/top-2 margin below the fp16 error).  A kernel that corrupts one tile in a hundred fails these."""
    B, H, W = (int(v) for v in G["r50_shape"])
    m, sd = _model(50, torch.float16)
    x = synth.pose_crops(SEED, B, H, W)
    hm = m(x.cuda()).cpu()
    want = pose_ref.pose_forward(sd, x, depth=50)
    err = (hm - want).abs().max().item()
    rng = (want.max() - want.min()).item()
    assert err <= 3e-3 * rng, f"fp16 heatmap error {err:.3e} vs range {rng:.2f}"
    oc, os_, oi = keypoints_ref.max_preds_ref(hm.numpy())
    same = oi == G["r50_idx"]
    assert same.mean() >= 0.97, f"only {same.mean():.3f} of arg-max indices match"
    margin = G["r50_margin"]
    assert np.all(margin[~same] <= 2 * err + 1e-6), "an arg-max flip that is not explained by a near-tie"
    # mAP@OKS with the CPU-reference keypoints as annotations
    center, scale = G["r50_center"], G["r50_scale"]
    coords, scores = evaluation.final_preds(hm.cuda(), center, scale, adjust_coords=True)
    pred = np.concatenate((coords, scores), axis=2)
    anno = np.concatenate((G["r50_final_coords_adjust1"], np.ones_like(scores)), axis=2)
    ap = evaluation.eval_mAP([pred], [anno], [scale * scale], evaluation.COCO_DELTA)
    print("fp16 max abs err", err, "argmax match", same.mean(), "AP@OKS", ap)
    assert ap[0] >= 0.99


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
def test_bn_batch_statistics(hip_lib, dtype):
    """nn.BatchNorm2d training-mode reduction (per-channel mean / biased variance over N*H*W) on NHWC."""
    from flowtrack.pytorch_amd.hip_ops import ActView, bn_batch_stats
    x = synth.normal(9, "bnx", (3, 72, 17, 13), std=2.0, mean=0.7)
    if dtype == torch.float16:
        x = x.half().float()
    buf = torch.zeros((3, 17, 13, 96), dtype=dtype, device="cuda")
    buf[..., :72] = x.permute(0, 2, 3, 1).to("cuda", dtype)
    mean, var = bn_batch_stats(ActView(buf, 72, 0))
    want_m = x.mean(dim=(0, 2, 3))
    want_v = x.var(dim=(0, 2, 3), unbiased=False)
    assert (mean.cpu() - want_m).abs().max() <= 1e-4 and (var.cpu() - want_v).abs().max() <= 1e-3


@pytest.mark.parametrize("adjust", [False, True])
def test_keypoints_inside_the_plan_match_the_separate_call(hip_lib, adjust):
    """forward_keypoints (max_preds recorded in the plan / graph) == forward + heatmap_max_preds."""
    from flowtrack.pytorch_amd.hip_ops import heatmap_max_preds
    m = models.deconv("resnet50", 17, False)
    m.load_state_dict(synth.fill_pose_state_dict(m.state_dict(), SEED))
    m = m.cuda().eval()
    m.compute_dtype = torch.float16
    x = synth.pose_crops(SEED + 9, 3).cuda()
    hm_ref = m(x)
    idx_r, score_r, coords_r = heatmap_max_preds(hm_ref, adjust_coords=adjust)
    m.keypoints_in_plan = adjust
    for _ in range(2):          # eager first call, then the graph replay
        hm, idx, score, coords = m.forward_keypoints(x)
        assert torch.equal(hm, hm_ref) and torch.equal(idx, idx_r) and torch.equal(score, score_r) and torch.equal(coords, coords_r)
        rows = m.forward_keypoint_rows(x)        # the same launch, read as (x, y, score) rows
        assert rows.is_contiguous() and torch.equal(rows, torch.cat((coords_r, score_r), dim=2))


@pytest.mark.parametrize("shape", [(2, 224, 160), (1, 288, 224), (3, 256, 192)], ids=str)
def test_cross_layer_fusions_do_not_change_the_network(hip_lib, shape):
    """fp16 plan with every cross-layer fusion (pooled stem, whole-bottleneck / head launches, K-concat shortcut, heatmap
    tail) vs the plan built from one launch per layer: the pooled stem and the t1 / t2 roundings are identical by
    construction, what remains is fp32 summation order — heatmaps agree to fp16 noise and the arg-max indices agree except
    near-ties; sizes include maps that are ragged for the 8x16 / 16x8 patches and the 8x8 pooled tiles."""
    B, H, W = shape
    fused, _ = _model(50, torch.float16)
    plain, _ = _model(50, torch.float16)
    plain.fuse_bottleneck = plain.fuse_stem_pool = plain.fuse_shortcut = plain.fuse_heatmap = False
    x = synth.pose_crops(SEED + 3, B, H, W).cuda()
    a, b = fused(x).float().cpu(), plain(x).float().cpu()
    assert {name for name, _ in fused._last_plan.prog.calls} >= {"ft_bottleneck_fwd"}
    assert "ft_bottleneck_fwd" not in {name for name, _ in plain._last_plan.prog.calls}
    rng = (b.max() - b.min()).item()
    assert (a - b).abs().max().item() <= 0.02 * rng
    ia, ib = a.flatten(2).argmax(2), b.flatten(2).argmax(2)
    flips = (ia != ib)
    if flips.any():      # every flip must be a near-tie in the one-launch-per-layer plan
        bf = b.flatten(2)
        gap = (bf.gather(2, ib.unsqueeze(2)) - bf.gather(2, ia.unsqueeze(2))).squeeze(2)[flips]
        assert gap.abs().max().item() <= 0.02 * rng
    assert flips.float().mean().item() <= 0.1


def test_stem_on_the_nchw_input_is_bit_identical_to_pack_plus_stem(hip_lib, monkeypatch):
    """fp16 plan whose fused stem gathers its patches from the NCHW fp32 input (ft_conv_desc.x_nchw_f32: no pack launch, no packed
    copy; resnet.py:19-23) vs the plan with ft_pack_nchw_to_nhwc in front of the same launch: with the tile picks shared, the
    heat maps are bit-identical (the same fp16 cast of the same pixels into the same LDS bytes)."""
    from flowtrack.pytorch_amd import hip_ops
    monkeypatch.setattr(hip_ops, "benchmark", False)          # the recorder's heuristics on both sides: identical launch lists
    x = synth.pose_crops(SEED + 5, 3, 256, 192).cuda()
    a, _ = _model(50, torch.float16)
    b, _ = _model(50, torch.float16)
    a.fuse_stem_pack, b.fuse_stem_pack = True, False
    ya, yb = a(x), b(x)
    na, nb = [n for n, _ in a._last_plan.prog.calls], [n for n, _ in b._last_plan.prog.calls]
    assert "ft_pack_nchw_to_nhwc" not in na and "ft_pack_nchw_to_nhwc" in nb, (na[:3], nb[:3])
    assert len(na) + 1 == len(nb), (len(na), len(nb))
    assert torch.equal(ya, yb), (ya.float() - yb.float()).abs().max().item()
    ra, rb = a(x), b(x)                                        # graph replays (the recorder's first option of every alternative)
    assert torch.equal(ra, rb), (ra.float() - rb.float()).abs().max().item()


def test_plan_alternatives_are_resolved_and_equivalent(hip_lib, monkeypatch):
    """A captured plan holds no unresolved choice (fused block vs three convs, direct vs implicit-GEMM 1x1): after the first
    call every alternative but one is gone, and forcing the OTHER option of every choice gives the same network to fp16
    summation-order noise (both forms of every piece are valid)."""
    from flowtrack.pytorch_amd import hip_ops
    x = synth.pose_crops(SEED + 21, 4).cuda()
    outs = []
    for flip in (False, True):
        monkeypatch.setattr(hip_ops, "_TILE_CACHE", {})
        monkeypatch.setattr(hip_ops, "_TILE_CACHE_LOADED", True)
        if flip:      # pre-seed the cache with the OTHER form of every choice the first model benchmarked (picks are names)
            forms = {"conv": ("direct", "igemm"), "bottleneck": ("fused", "convs"), "entry": ("fused", "head+exit"),
                     "head2": ("convs", "fused")}
            for k, picked in seen.items():
                a, b = forms[k.split("|")[1]]
                hip_ops._TILE_CACHE[k] = b if picked == a else a
        m, _ = _model(50, torch.float16)
        y = m(x)
        y2 = m(x)       # graph replay
        assert torch.equal(y, y2)
        names = [name for name, _ in m._last_plan.prog.calls]
        assert not any(n in ("__choice__", "__option__", "__endchoice__") for n in names)
        seen = {k: v for k, v in hip_ops._TILE_CACHE.items() if k.startswith("choice|")}
        assert seen, "the plan recorded no alternative at all"
        outs.append((y.float().cpu(), names))
    (a, na), (b, nb) = outs
    assert na != nb
    rng = (a.max() - a.min()).item()
    assert (a - b).abs().max().item() <= 0.02 * rng


def test_in_place_parameter_edits_rebuild_the_packed_weights(hip_lib):
    """Plans hold packed copies of the parameters: an edit that bypasses the model's load_state_dict — an in-place op on the
    parameter (bumps its version counter), a load through a child module — must not keep serving the old weights, and ordinary
    forwards must not rebuild anything.  Edits through `.data` carry no version counter and are NOT detected: refresh() is the
    contract for those (engine.py), checked at the end."""
    m, _ = _model(50, torch.float16)
    x = synth.pose_crops(SEED + 5, 2).cuda()
    a = m(x)
    plan = m._last_plan
    assert torch.equal(m(x), a) and m._last_plan is plan and plan.runs == 2      # no rebuild between plain forwards
    with torch.no_grad():
        m.heatmap.bias.add_(0.5)                       # in place, no load_state_dict (nn.init.* does the same)
    b = m(x)
    assert m._last_plan is not plan
    assert (b - a - 0.5).abs().max().item() <= 2e-3
    sd = {k: v.clone() for k, v in m.heatmap.state_dict().items()}
    sd["bias"] -= 0.5
    m.heatmap.load_state_dict(sd)                      # through a child module
    assert (m(x) - a).abs().max().item() <= 2e-3
    m.heatmap.bias.data.add_(0.25)                     # `.data` edits carry no version counter: refresh() is the contract
    m.refresh()
    assert (m(x) - a - 0.25).abs().max().item() <= 2e-3


def test_row_gatherer_comm_stream_path_on_one_gpu(hip_lib):
    """The branch of parallel.RowGatherer that only world > 1 on GPUs takes — wait_stream on the compute stream, the exchange
    on a communication stream, an event handed back, finish() one step late — run on ONE GPU (force_stream: a device copy
    stands in for the collective) while the pose plan's graph replays on the compute stream: over 50 steps with a different
    batch each, the lagged rows are exactly the rows step t-1 produced.  (Replaces, on the reference side, the DataParallel
    gather of tools/flownet/main.py:133-134,186,197.)"""
    from flowtrack.pytorch_amd import parallel
    B = 8
    m, _ = _model(50, torch.float16)
    m.keypoints_in_plan = True
    x = m.static_input(B, 256, 192)
    base = synth.pose_crops(SEED + 31, B).cuda()
    g = parallel.RowGatherer(B, (17, 3), torch.float32, x.device, depth=2, force_stream=True)
    assert g.stream is not None and g.world == 1
    produced, lagged, pending = [], [], None
    for t in range(50):
        x.copy_(torch.roll(base, shifts=3 * t, dims=3))
        rows = m.forward_keypoint_rows(x)
        produced.append(rows.clone())
        h = g.start(rows)
        if pending is not None:
            lagged.append(g.finish(pending).clone())
        pending = h
    lagged.append(g.finish(pending).clone())
    torch.cuda.synchronize()
    assert len(lagged) == 50 and all(torch.equal(a, b) for a, b in zip(produced, lagged))
    assert not all(torch.equal(produced[0], p) for p in produced[1:]), "the batches were meant to differ"
    # more starts outstanding than slots: the slot guard makes the refill wait for the exchange that still owns the slot
    hs = [g.start(produced[i]) for i in range(5)]
    torch.cuda.synchronize()
    assert torch.equal(g.finish(hs[-1]), produced[4]) and torch.equal(g.finish(hs[-2]), produced[3])


def test_bench_runners_take_the_comm_stream_path_on_one_gpu(hip_lib):
    """bench.py's N > 1 step (graph replay -> RowGatherer.start -> consume the previous step's rows) for both workloads, on
    one GPU through the force_gather hook: the drained result equals a plain forward of the same resident batch."""
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda", 0)
    args = types.SimpleNamespace()
    model, x, step = bench.make_pose_runner(args, dev, torch.float16, 0, 1, "resnet50", 256, 192, 4, force_gather=True)
    for _ in range(6):
        rows = step()
    step.drain()
    torch.cuda.synchronize()
    want = model.forward_keypoint_rows(x).clone()
    torch.cuda.synchronize()
    assert torch.equal(rows, want) and torch.isfinite(want).all()
    fmodel, fx, fstep = bench.make_flow_runner(args, dev, torch.float16, 0, 1, "FlowNet2S", 2, "sampled", force_gather=True)
    for _ in range(4):
        flow = fstep()
    fstep.drain()
    torch.cuda.synchronize()
    assert tuple(flow.shape) == (2, 2, 384, 512) and torch.isfinite(flow).all()


@pytest.mark.parametrize("pair", [(64, 128), (8, 256)], ids=["b64_then_b128", "b8_then_b256"])
def test_two_batch_sizes_of_one_model_get_their_own_conv_direct_streams(hip_lib, monkeypatch, pair):
    """ft_conv_direct_fwd's kernel form follows the pixel count (K split 1 / 4, weight-stationary), every form orders the
    weight stream differently and all have the same byte count: two plans of ONE model on either side of a form boundary
    (tracking's pose_est_frames builds plans for buckets 8..256) must each run on a stream of their own layout
    (ft_conv_direct_stream_id).  Checked against a model whose 1x1 convs stay on ft_conv2d_fwd, heuristic picks on both
    sides (no benchmark: the direct form is the recorder's first choice wherever it exists)."""
    from flowtrack.pytorch_amd import hip_ops
    monkeypatch.setattr(hip_ops, "benchmark", False)
    monkeypatch.setattr(hip_ops, "_TILE_CACHE", {})
    monkeypatch.setattr(hip_ops, "_TILE_CACHE_LOADED", True)
    m, _ = _model(50, torch.float16)
    monkeypatch.setattr(hip_ops, "CONV_DIRECT", False)
    ref, _ = _model(50, torch.float16)
    seen_ids = {}
    for B in pair:
        x = synth.pose_crops(SEED + 40 + B, B).cuda()
        monkeypatch.setattr(hip_ops, "CONV_DIRECT", True)
        got = m(x).float().cpu()
        names = [n for n, _ in m._last_plan.prog.calls]
        assert "ft_conv_direct_fwd" in names, "the plan under test runs no ft_conv_direct_fwd launch at all"
        for rec in m._last_plan.prog.conv_records:
            if names[rec[1]] == "ft_conv_direct_fwd":
                import ctypes
                seen_ids.setdefault(rec[0], set()).add(int(hip_lib.ft_conv_direct_stream_id(ctypes.byref(rec[3]))))
        monkeypatch.setattr(hip_ops, "CONV_DIRECT", False)
        want = ref(x).float().cpu()
        rng = (want.max() - want.min()).item()
        err = (got - want).abs().max().item()
        assert err <= 0.02 * rng, f"batch {B}: heat maps differ from the ft_conv2d_fwd path by {err:.3e} (range {rng:.2f})"
    assert any(len(v) > 1 for v in seen_ids.values()), "no layer changed its stream layout between the two batch sizes: the pair does not test the key"


def test_min_margin_kernel_matches_topk(hip_lib):
    """ft_heatmap_min_margin: per crop the smallest (largest - second largest) over its maps; ties -> 0."""
    import ctypes
    from flowtrack.pytorch_amd.hip_ops import current_stream_handle
    hm = synth.normal(77, "margin.hm", (5, 17, 64, 48)).cuda().contiguous()
    hm[1, 3, 10, 7] = hm[1, 3].max() + 0.25        # a clear winner ...
    hm[2, 5, 0, 0] = hm[2, 5, 63, 47] = hm[2, 5].max() + 1.0     # ... and an exact tie in another crop
    out = torch.empty(5, dtype=torch.float32, device="cuda")
    assert hip_lib.ft_heatmap_min_margin(hm.data_ptr(), 5, 17, 64, 48, out.data_ptr(), current_stream_handle()) == 0
    top2 = hm.flatten(2).topk(2, dim=2).values
    want = (top2[..., 0] - top2[..., 1]).min(dim=1).values
    assert torch.equal(out, want) and out[2].item() == 0.0


def test_argmax_screen_is_selective(hip_lib):
    """ft_heatmap_argmax_screen on synthetic maps of the two regimes.  Single-peak maps (a Gaussian bump of sigma 2 px centred
    on a pixel, amplitude ~1, plus noise of 1e-3: top-1 / top-2 margin ~0.1 of the range, what a trained pose net emits,
    lib/pose/utils/evaluation.py:11-20 reads their arg-max): nothing is flagged.  The cases the screen exists for are each
    flagged: a near-tie (margin < 2 E), a maximum within E of zero (the `score > 0` mask), a NaN, an Inf.  Noise-like maps:
    (nearly) all flagged.  The statistics row is (smallest margin, range, smallest |top-1|, E)."""
    import ctypes
    from flowtrack.pytorch_amd.hip_ops import current_stream_handle
    N, K, H, W = 40, 17, 64, 48
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    cy = (4 + (synth.uniform01(91, "cy", (N, K)) * (H - 8))).astype(np.int64)
    cx = (4 + (synth.uniform01(91, "cx", (N, K)) * (W - 8))).astype(np.int64)
    amp = torch.from_numpy(0.7 + 0.3 * synth.uniform01(91, "amp", (N, K)).astype(np.float32))
    hm = amp[..., None, None] * torch.exp(-((yy - torch.from_numpy(cy)[..., None, None]) ** 2 + (xx - torch.from_numpy(cx)[..., None, None]) ** 2) / 8.0)
    hm = (hm + 1e-3 * synth.normal(91, "noise", (N, K, H, W))).contiguous()
    rel = 1.6e-3
    special = {3: "tie", 7: "zero", 11: "nan", 13: "inf"}
    hm[3, 2, cy[3, 2], cx[3, 2] + 1] = hm[3, 2, cy[3, 2], cx[3, 2]] - 1e-3         # a neighbour within 2 E of the peak
    hm[7, 5] = hm[7, 5] - hm[7, 5].max() + 5e-4                                     # this map's maximum sits at +5e-4 < E
    hm[11, 0, 1, 1] = float("nan")
    hm[13, 16, 2, 2] = float("inf")
    g = hm.cuda()
    flags = torch.empty(N, dtype=torch.int32, device="cuda")
    stats = torch.empty((N, 4), dtype=torch.float32, device="cuda")
    assert hip_lib.ft_heatmap_argmax_screen(g.data_ptr(), N, K, H, W, ctypes.c_float(rel), flags.data_ptr(), stats.data_ptr(),
                                            current_stream_handle()) == 0
    f, st = flags.cpu().numpy(), stats.cpu().numpy()
    assert sorted(np.nonzero(f)[0].tolist()) == sorted(special), f"flagged {np.nonzero(f)[0].tolist()}"
    clean = [n for n in range(N) if n not in special]
    top2 = hm[clean].flatten(2).topk(2, dim=2).values
    assert np.allclose(st[clean, 0], (top2[..., 0] - top2[..., 1]).min(dim=1).values.numpy(), rtol=0, atol=1e-6)
    rng = hm[clean].flatten(1).max(dim=1).values - hm[clean].flatten(1).min(dim=1).values
    assert np.allclose(st[clean, 1], rng.numpy(), atol=1e-6) and np.allclose(st[clean, 3], rel * rng.numpy(), atol=1e-7)
    assert (st[clean, 0] > 20 * st[clean, 3]).all(), "single-peak maps: margins are an order of magnitude above the bound"
    noise = synth.normal(92, "noise.hm", (N, K, H, W)).cuda().contiguous()
    assert hip_lib.ft_heatmap_argmax_screen(noise.data_ptr(), N, K, H, W, ctypes.c_float(rel), flags.data_ptr(), stats.data_ptr(),
                                            current_stream_handle()) == 0
    assert flags.float().mean().item() >= 0.5, "noise-like maps: the smallest of 17 top-1 / top-2 gaps is inside the bound for most crops"
    assert hip_lib.ft_heatmap_argmax_screen(noise.data_ptr(), N, 257, 8, 6, ctypes.c_float(rel), flags.data_ptr(), stats.data_ptr(),
                                            current_stream_handle()) != 0       # K > 256: unsupported, said so


def test_fp16_exact_argmax_mode_has_the_fp32_argmax(hip_lib):
    """DeconvResnet.forward_keypoint_rows_exact: fp16 pass, margin screen, fp32 re-run of the screened crops — every key point's
    arg-max pixel equals the fp32 parity mode's (which equals the CPU reference's: test_pose_fp32_matches_reference_golden) on
    192 synthetic crops; some crops are re-run, far from all of them would be a useless screen, and crops that are not re-run
    keep the fp16 rows."""
    m16, _ = _model(50, torch.float16)
    m32, _ = _model(50, torch.float32)
    m16.keypoints_in_plan = m32.keypoints_in_plan = True
    W = 48
    idx = lambda rows: (torch.floor(rows[..., 1] + 0.5) * W + torch.floor(rows[..., 0] + 0.5)).long()
    reran = flips16 = 0
    for k in range(3):
        x = synth.pose_crops(SEED + 60 + k, 64).cuda()
        plain = m16.forward_keypoint_rows(x).clone()
        rows, n = m16.forward_keypoint_rows_exact(x)
        want = m32.forward_keypoint_rows(x).clone()
        assert torch.equal(idx(rows), idx(want)), "exact mode: an arg-max differs from the fp32 parity mode"
        flips16 += int((idx(plain) != idx(want)).sum())
        reran += n
        untouched = (rows == plain).flatten(1).all(dim=1)
        assert int((~untouched).sum()) <= n
    assert 0 < reran <= 192, f"{reran} of 192 crops re-run"       # noise-like maps (random weights): nearly all of them
    print("fp16 arg-max flips without the screen:", flips16, "of", 192 * 17, "- crops re-run:", reran)


def test_exact_argmax_submit_finish_equals_the_blocking_form(hip_lib):
    """DeconvResnet.exact_submit / exact_finish (round 5: screen -> flag compaction -> gather of the flagged crops, all on the
    device; the host looks at the count one step later) give the rows of forward_keypoint_rows_exact bit for bit, for a bound that
    flags nothing, one that flags about half of the crops and the default (noise-like maps: nearly all); two steps are kept in
    flight and finished late, with the input buffer overwritten in between; ft_gather_flagged_rows itself is checked on a
    planted flag vector."""
    m16, _ = _model(50, torch.float16)
    m16.keypoints_in_plan = True
    xa, xb = synth.pose_crops(SEED + 70, 64).cuda(), synth.pose_crops(SEED + 71, 64).cuda()
    # a bound that splits the batch: the median of the per-crop ratio (smallest margin / (2 R))
    m16.forward_keypoint_rows_exact(xa)
    st = m16._last_screen_stats.cpu()
    mid = float((st[:, 0] / (2 * st[:, 1])).median())
    keep = m16.exact_argmax_rel_bound
    try:
        for bound, lo, hi in ((0.0, 0, 0), (mid, 8, 56), (keep, 32, 64)):
            m16.exact_argmax_rel_bound = bound
            want_a, na = m16.forward_keypoint_rows_exact(xa)
            want_b, nb = m16.forward_keypoint_rows_exact(xb)
            want_a, want_b = want_a.clone(), want_b.clone()
            assert lo <= na <= hi, (bound, na)
            x = xa.clone()
            ha = m16.exact_submit(x)
            x.copy_(xb)                                  # the caller's buffer is free as soon as submit returns
            hb = m16.exact_submit(x)
            x.zero_()
            with pytest.raises(Exception):
                m16.exact_submit(x)                      # both staging slots in flight
            rows_a, ga = m16.exact_finish(ha)
            rows_b, gb = m16.exact_finish(hb)
            assert (ga, gb) == (na, nb)
            assert torch.equal(rows_a, want_a) and torch.equal(rows_b, want_b), bound
            with pytest.raises(Exception):
                m16.exact_finish(ha)
    finally:
        m16.exact_argmax_rel_bound = keep
    # the same with the device side recorded INSIDE the plan's graph (exact_in_plan): plan replicas in rotation, finished one step late
    m16.exact_in_plan = True
    try:
        for bound, lo, hi in ((0.0, 0, 0), (mid, 8, 56)):
            m16.exact_argmax_rel_bound = bound
            m16.exact_in_plan = False
            want_a, na = m16.forward_keypoint_rows_exact(xa)
            want_b, nb = m16.forward_keypoint_rows_exact(xb)
            want_a, want_b = want_a.clone(), want_b.clone()
            m16.exact_in_plan = True
            pa, pb = m16.plan_for(64, 256, 192, 0), m16.plan_for(64, 256, 192, 1)
            assert pa is not pb and pa.exact["rel_bound"] == bound
            for rep in range(2):                         # second pass: the captured graph
                pa.x_static.copy_(xa)
                pb.x_static.copy_(xb)
                m16.exact_submit_plan(pa)
                m16.exact_submit_plan(pb)
                with pytest.raises(Exception):
                    m16.exact_submit_plan(pa)            # not finished yet
                ra, ga = m16.exact_finish_plan(pa)
                rb, gb = m16.exact_finish_plan(pb)
                assert (ga, gb) == (na, nb), (bound, rep, ga, gb, na, nb)
                assert torch.equal(ra, want_a) and torch.equal(rb, want_b), (bound, rep)
    finally:
        m16.exact_in_plan = False
        m16.exact_argmax_rel_bound = keep
    # the gather alone: rows 3, 4, 17, 63 of 64 flagged
    import ctypes
    from flowtrack.pytorch_amd.hip_ops import current_stream_handle
    flags = torch.zeros(64, dtype=torch.int32, device="cuda")
    flags[[3, 4, 17, 63]] = 1
    src = torch.arange(64 * 48, dtype=torch.float32, device="cuda").reshape(64, 48)
    dst = torch.full((64, 48), -1.0, device="cuda")
    hdr = torch.full((65,), -7, dtype=torch.int32, device="cuda")
    assert hip_lib.ft_gather_flagged_rows(flags.data_ptr(), 64, src.data_ptr(), 48 * 4, dst.data_ptr(), hdr.data_ptr(), current_stream_handle()) == 0
    torch.cuda.synchronize()
    assert hdr[:5].tolist() == [4, 3, 4, 17, 63] and int(hdr[5]) == -7
    assert torch.equal(dst[:4], src[[3, 4, 17, 63]]) and float(dst[4:].max()) == -1.0
    flags.zero_()
    assert hip_lib.ft_gather_flagged_rows(flags.data_ptr(), 64, src.data_ptr(), 48 * 4, dst.data_ptr(), hdr.data_ptr(), current_stream_handle()) == 0
    torch.cuda.synchronize()
    assert int(hdr[0]) == 0
    assert hip_lib.ft_gather_flagged_rows(flags.data_ptr(), 2000, src.data_ptr(), 48 * 4, dst.data_ptr(), hdr.data_ptr(), current_stream_handle()) != 0
