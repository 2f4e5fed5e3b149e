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
    oracle. Bar: max-abs <= 5e-2 of the heatmap range, >= 90 % identical arg-max, and every mismatch
    must be a near-tie (reference top-1/top-2 margin below the fp16 error)."""
    B, H, W = (int(v) for v in G["r50_shape"])
    m, sd = _model(50, torch.float16)
    x = synth.pose_crops(SEED, B, H, W)
    hm = m(x.cuda()).cpu()
    want = pose_ref.pose_forward(sd, x, depth=50)
    err = (hm - want).abs().max().item()
    rng = (want.max() - want.min()).item()
    assert err <= 5e-2 * rng, f"fp16 heatmap error {err:.3e} vs range {rng:.2f}"
    oc, os_, oi = keypoints_ref.max_preds_ref(hm.numpy())
    same = oi == G["r50_idx"]
    assert same.mean() >= 0.9, f"only {same.mean():.3f} of arg-max indices match"
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
            forms = {"conv": ("direct", "igemm"), "bottleneck": ("fused", "convs"), "entry": ("fused", "head+exit")}
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
