"""Dev: where the sequential pass of the 300-frame clip spends its time (cProfile on rank 0's tracking_pass)."""
import cProfile, os, pstats, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tools.tracking import demo

args = types.SimpleNamespace(pose_backbone=50, pose_model="", flow_net="FlowNet2S", flow_model="", fp16=True)
dev = torch.device("cuda", 0)
pose, flow = demo.build_nets(args, dev)
frames, dets = demo.synthetic_clip(300)
demo.run_clip(frames, dets, pose, flow, max_boxes="2x")
t0 = time.perf_counter(); out, tm = demo.run_clip(frames, dets, pose, flow, max_boxes="2x"); dt = time.perf_counter() - t0
print(f"clip {dt:.3f} s = {300 / dt:.1f} frames/s", tm)
pr = cProfile.Profile(); pr.enable(); demo.run_clip(frames, dets, pose, flow, max_boxes="2x"); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
