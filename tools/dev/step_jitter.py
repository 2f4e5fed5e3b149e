"""Dev: per-step time distribution of one workload (looks for bimodal behaviour between processes / steps)."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from flowtrack.pytorch_amd import synth

w = sys.argv[1] if len(sys.argv) > 1 else "flow"
dev = torch.device("cuda:0")
if w == "flow":
    model = bench.build_flow(dev, torch.float16); x = synth.frame_pairs(100, 16).to(dev)
else:
    model = bench.build_pose(dev, torch.float16); x = synth.pose_crops(100, 64).to(dev)
for _ in range(3):
    model(x, copy_output=False)
torch.cuda.synchronize()
ts = []
for i in range(60):
    t0 = time.perf_counter(); model(x, copy_output=False); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter()
for i in range(30):
    model(x, copy_output=False)
torch.cuda.synchronize()
pipelined = (time.perf_counter() - t0) / 30 * 1e3
s = sorted(ts)
print(f"{w}: sync'd steps min {s[0]:.3f} med {s[30]:.3f} p90 {s[54]:.3f} max {s[-1]:.3f} ms; pipelined {pipelined:.3f} ms/step; first10 " + " ".join(f"{t:.2f}" for t in ts[:10]))
