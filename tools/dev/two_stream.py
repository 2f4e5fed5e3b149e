"""Experiment: one B=64 plan vs two B=32 plans on two streams (overlapping kernel tails/prologues)."""
import sys, os, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.pose import models
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 64
ms = []
for i in range(nsplit):
    m = models.deconv("resnet50", 17, False); m.load_state_dict(synth.fill_pose_state_dict(m.state_dict(), 1))
    m = m.cuda().eval(); m.compute_dtype = torch.float16
    ms.append(m)
xs = [synth.pose_crops(1, B // nsplit).cuda() for _ in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
def step():
    for m, x, s in zip(ms, xs, streams):
        with torch.cuda.stream(s):
            m(x, copy_output=False)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
print(f"nsplit={nsplit}: {dt*1e3:.3f} ms/step -> {B/dt:.0f} crops/s")
