"""Parity report of the pose path (BASELINE.json metric: "mAP@OKS vs CPU ref"): GPU heatmaps / keypoints in fp32 (parity
mode) and fp16 (fast mode) against the CPU oracle on the same synthetic crops, with the oracle's keypoints as annotations
(SURVEY §8(d)).  Not a pytest module — test infrastructure that needs the GPU and the oracle:

    python tests/parity_report.py [--crops 64] [--backbone resnet50] > profiles/r01_pose_parity_report.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flowtrack.pytorch_amd import synth                      # noqa: E402
from flowtrack.pytorch_amd.pose import evaluation, models   # noqa: E402
from oracle import flow_ref, keypoints_ref, pose_ref         # noqa: E402


def flow_report(pairs, seed):
    """FlowNet2S / FlowNet2C / FlowNet2CS at 512x384 (BASELINE configs[3] shape): GPU fp32 and fp16 flow vs the CPU oracle."""
    import types
    from flowtrack.pytorch_amd.flownet import models as fmodels
    x = synth.frame_pairs(seed + 1, pairs, 384, 512)
    out = {"pairs": pairs, "res": "512x384", "seed": seed, "models": {}}
    for name, ref in (("FlowNet2S", flow_ref.flownet2s_forward), ("FlowNet2C", flow_ref.flownet2c_forward),
                      ("FlowNet2CS", flow_ref.flownet2cs_forward)):
        m = getattr(fmodels, name)(types.SimpleNamespace(rgb_max=255.0, fp16=False))
        sd = synth.fill_flow_state_dict(m.state_dict(), seed)
        m.load_state_dict(sd)
        m = m.cuda().eval()
        want = torch.cat([ref(sd, x[i:i + 1]) for i in range(pairs)])
        mag = want.norm(dim=1).mean().item()
        row = {"mean_flow_magnitude_px": mag}
        for mode, dtype in (("fp32", torch.float32), ("fp16", torch.float16)):
            m.compute_dtype = dtype
            got = m(x.cuda()).float().cpu()
            row[mode] = {"max_abs_err_px": float((got - want).abs().max()), "EPE_px": float((got - want).norm(dim=1).mean())}
        out["models"][name] = row
    print(json.dumps(out, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["pose", "flow"], default="pose")
    ap.add_argument("--pairs", type=int, default=4)
    ap.add_argument("--crops", type=int, default=64)
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--res", default="256x192")
    ap.add_argument("--seed", type=int, default=2024)
    a = ap.parse_args()
    if a.workload == "flow":
        return flow_report(a.pairs, a.seed)
    H, W = (int(v) for v in a.res.split("x"))
    depth = int(a.backbone[len("resnet"):])
    m = models.deconv(a.backbone, 17, False)
    sd = synth.fill_pose_state_dict(m.state_dict(), a.seed)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    x = synth.pose_crops(a.seed + 1, a.crops, H, W)
    want = torch.cat([pose_ref.pose_forward(sd, x[i:i + 4], depth=depth) for i in range(0, a.crops, 4)])
    center = np.stack((np.full(a.crops, W / 2.0), np.full(a.crops, H / 2.0)), 1)
    scale = np.full(a.crops, H / 200.0)
    ref_coords, ref_scores, ref_idx, _ = keypoints_ref.final_preds_ref(want.numpy(), center, scale, adjust_coords=True)
    anno = np.concatenate((ref_coords, np.ones_like(ref_scores)), axis=2)
    rng = (want.max() - want.min()).item()
    out = {"backbone": a.backbone, "crops": a.crops, "res": a.res, "seed": a.seed, "heatmap_range": rng,
           "annotations": "CPU-oracle keypoints (final_preds, adjust_coords) of the same crops", "modes": {}}
    for name, dtype in (("fp32", torch.float32), ("fp16", torch.float16)):
        m.compute_dtype = dtype
        hm = m(x.cuda())
        coords, scores = evaluation.final_preds(hm, center, scale, adjust_coords=True)
        _, _, idx = keypoints_ref.max_preds_ref(hm.cpu().numpy())
        pred = np.concatenate((coords, scores), axis=2)
        # ref_scale = box area in the reference's scale units (1 = 200 px), as tests/test_pose_gpu.py: stricter than px^2
        aps = evaluation.eval_mAP([pred], [anno], [scale * scale], evaluation.COCO_DELTA)
        out["modes"][name] = {
            "heatmap_max_abs_err": float((hm.cpu() - want).abs().max()),
            "argmax_identical_frac": float((idx == ref_idx).mean()),
            "keypoint_max_abs_err_px": float(np.abs(coords - ref_coords).max()),
            "AP_at_OKS_0.50_to_0.95": [float(v) for v in np.atleast_1d(aps)],
            "mAP_at_OKS": float(np.mean(aps)),
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
