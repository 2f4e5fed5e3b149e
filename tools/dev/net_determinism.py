"""Dev: replay a whole plan many times on the same resident batch and compare the output bits of every replay with the first
(catches rare races: hand-counted waits, LDS-DMA ordering, scratch aliasing).  usage: net_determinism.py [replays] [cold]
cold (round 5): 512 MB of unrelated traffic before every replay, so the weights and the first layers' inputs arrive from HBM — the
condition under which the first version of bottleneck_cluster_kernel read an LDS-DMA chunk before it had landed."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cold = len(sys.argv) > 2 and sys.argv[2] == "cold"
junk = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device="cuda") if cold else None
bad = 0
for name, B, H, W in (("resnet50", 64, 256, 192), ("resnet101", 16, 384, 288), ("resnet50", 8, 256, 192), ("FlowNet2S", 16, 384, 512), ("FlowNet2C", 4, 384, 512)):
    if name.startswith("resnet"):
        from flowtrack.pytorch_amd.pose import models
        m = models.deconv(name, 17, False); m.load_state_dict(synth.fill_pose_state_dict(m.state_dict(), 1))
        x = synth.pose_crops(1, B, H, W)
    else:
        from flowtrack.pytorch_amd.flownet import models
        m = getattr(models, name)(types.SimpleNamespace(rgb_max=255.0, fp16=False)); m.load_state_dict(synth.fill_flow_state_dict(m.state_dict(), 1))
        x = synth.frame_pairs(1, B, H, W)
    m = m.cuda().eval(); m.compute_dtype = torch.float16
    x = x.cuda()
    for _ in range(3):
        ref = m(x).clone()
    torch.cuda.synchronize()
    diff = 0
    for i in range(reps):
        if cold:
            junk.add_(1)
        y = m(x, copy_output=False)
        if not torch.equal(y, ref):
            diff += 1
    torch.cuda.synchronize()
    print(f"{name} B={B} {H}x{W}{' (cold caches)' if cold else ''}: {diff} of {reps} replays differ from the first", flush=True)
    bad += diff
    del m
sys.exit(1 if bad else 0)
