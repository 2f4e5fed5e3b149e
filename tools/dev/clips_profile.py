"""cProfile of tools/tracking/demo.run_clips (K clips interleaved on one GPU): where the host thread's time goes."""
import cProfile, os, pstats, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tools.tracking import demo
K, T = int(sys.argv[1]) if len(sys.argv) > 1 else 6, int(sys.argv[2]) if len(sys.argv) > 2 else 150
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
args = types.SimpleNamespace(pose_backbone=50, pose_model="", flow_net="FlowNet2S", flow_model="", fp16=True)
dev = torch.device("cuda", 0)
pose, flow = demo.build_nets(args, dev)
clips = [demo.synthetic_clip(T, seed=c) for c in range(K)]
demo.run_clips(clips, pose, flow, max_boxes="2x", groups=G)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
demo.run_clips(clips, pose, flow, max_boxes="2x", groups=G)
pr.disable()
print(f"{K} x {T} frames: {time.perf_counter() - t0:.3f} s under cProfile")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
