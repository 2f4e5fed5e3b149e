"""Where does the fp16 mode's heat-map error come from?  (run on the GPU box:  python tests/error_budget.py)

For every stage s of the ResNet-50 pose net: the fp16 plan's activation behind s (as the GPU wrote it), relative to the fp32
plan's, and — the budget proper — the heat maps one gets when everything THROUGH s is the fp16 mode's and everything after it is
exact (the CPU oracle continued from that activation, oracle/pose_ref.py `start=`): their error against the all-fp32 heat maps and
the arg-max flips they cause.  Row s minus row s-1 is what stage s adds.  The last row is the complete fp16 mode (with its hi / lo
fp16 heat-map weights and fp32 accumulation in the fused tail).  Checker role only: nothing here is on the product path."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flowtrack.pytorch_amd import synth  # noqa: E402
from flowtrack.pytorch_amd.pose import models  # noqa: E402
from oracle import keypoints_ref, pose_ref  # noqa: E402


def main():
    B, H, W, seed = 64, 256, 192, 1234
    dev = torch.device("cuda:0")
    nets = {}
    for name, dt in (("fp32", torch.float32), ("fp16", torch.float16)):
        m = models.deconv("resnet50", 17, False)
        m.load_state_dict(synth.fill_pose_state_dict(m.state_dict(), seed))
        m = m.to(dev).eval()
        m.compute_dtype = dt
        nets[name] = m
    sd = {k: v.detach().float().cpu() for k, v in nets["fp32"].state_dict().items()}
    x = synth.pose_crops(100, B, H, W)                       # the benchmarked batch (bench.py, rank 0)
    hm = {k: m(x.to(dev)).float().cpu() for k, m in nets.items()}
    plans = {k: m._last_plan for k, m in nets.items()}
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    want = hm["fp32"]
    _, _, ref_idx = keypoints_ref.max_preds_ref(want.numpy())
    rng = float(want.max() - want.min())

    def nchw(view):
        return view.t[..., view.coff:view.coff + view.C].permute(0, 3, 1, 2).float().cpu().contiguous()

    rows = []
    for s in ("stem", "layer1", "layer2", "layer3", "layer4", "deconv.0", "deconv.3"):
        a16, a32 = nchw(plans["fp16"].stages[s]), nchw(plans["fp32"].stages[s])
        act_rng = float(a32.max() - a32.min())
        cont = torch.cat([pose_ref.pose_forward(sd, a16[i:i + 8], start=s) for i in range(0, B, 8)])
        check = pose_ref.pose_forward(sd, a32[:4], start=s)          # sanity: continuing the fp32 activation reproduces the fp32 maps
        assert float((check - want[:4]).abs().max()) <= 1e-3, s
        _, _, idx = keypoints_ref.max_preds_ref(cont.numpy())
        rows.append({"fp16_through": s, "activation_max_abs_err": float((a16 - a32).abs().max()), "activation_range": act_rng,
                     "activation_rel_rms_err": float(((a16 - a32).pow(2).mean().sqrt() / a32.pow(2).mean().sqrt())),
                     "heatmap_max_abs_err": float((cont - want).abs().max()), "heatmap_rms_err": float((cont - want).pow(2).mean().sqrt()),
                     "argmax_flips": int((idx != ref_idx).sum())})
    _, _, idx16 = keypoints_ref.max_preds_ref(hm["fp16"].numpy())
    rows.append({"fp16_through": "deconv.6 + heatmap (the whole fp16 mode)", "heatmap_max_abs_err": float((hm["fp16"] - want).abs().max()),
                 "heatmap_rms_err": float((hm["fp16"] - want).pow(2).mean().sqrt()), "argmax_flips": int((idx16 != ref_idx).sum())})
    out = {"crops": B, "keypoints": int(ref_idx.size), "heatmap_range": rng, "rows": rows}
    print(json.dumps(out, indent=1))
    for r in rows:
        print(f"  {r['fp16_through']:44s} heat-map err max {r['heatmap_max_abs_err']:.3e} rms {r['heatmap_rms_err']:.3e}  flips {r['argmax_flips']:3d}"
              + (f"   activation rel rms {r['activation_rel_rms_err']:.2e}" if "activation_rel_rms_err" in r else ""), file=sys.stderr)


if __name__ == "__main__":
    main()
