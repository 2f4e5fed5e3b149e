"""Per-launch timing of a whole plan (dev tool): python tools/dev/net_bench.py FlowNet2C 16 384 512 fp16"""
import sys, os, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
name, B, H, W, dt = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
dtype = torch.float16 if dt == "fp16" else torch.float32
if name.startswith("resnet"):
    from flowtrack.pytorch_amd.pose import models
    m = models.deconv(name, 17, False); m.load_state_dict(synth.fill_pose_state_dict(m.state_dict(), 1))
    x = synth.pose_crops(1, B, H, W)
else:
    from flowtrack.pytorch_amd.flownet import models
    m = getattr(models, name)(types.SimpleNamespace(rgb_max=255.0, fp16=False)); m.load_state_dict(synth.fill_flow_state_dict(m.state_dict(), 1))
    x = synth.frame_pairs(1, B, H, W)
m = m.cuda().eval(); m.compute_dtype = dtype
x = x.cuda()
for _ in range(3): m(x)
torch.cuda.synchronize()
plan = next(iter(m._plans.values()))
times = plan.prog.time_calls(iters=5)
hot = dict(enumerate(t for _, t in plan.prog.time_calls(iters=5, repeat_hot=True))) if os.environ.get("NB_HOT") else {}
labels = {idx: lab for lab, idx, _, _ in plan.prog.conv_records}
labels.update({r[1]: r[0] for r in plan.prog.fused_records})
flops = {r[1]: r[2] for r in list(plan.prog.conv_records) + list(plan.prog.fused_records)}
tot = 0.0
for i, (nm, ms) in enumerate(times):
    tot += ms
    if nm.startswith("__"): continue
    print(f"{i:3d} {nm:28s} {labels.get(i, ''):28s} {ms*1e3:9.1f} us" + (f"  {flops[i] / ms * 1e-9:7.1f} TF/s" if i in flops else "")
          + (f"   hot {hot[i]*1e3:7.1f} us  (cold - hot {1e3 * (ms - hot[i]):+6.1f})" if hot else ""))
print(f"sum of launches {tot:.3f} ms  -> {B/tot*1e3:.0f} units/s (eager, event-timed)")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m(x, copy_output=False)
torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 20
print(f"graph replay {dt_*1e3:.3f} ms/step -> {B/dt_:.0f} units/s")
