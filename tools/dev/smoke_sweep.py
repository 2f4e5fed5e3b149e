"""Dev: run every model family at a few odd batch sizes / resolutions / dtypes; checks finiteness and fp16-vs-fp32 agreement."""
import sys, os, types, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.pose import models as pm
from flowtrack.pytorch_amd.flownet import models as fm

def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()

ok = True
for backbone, nc, B, H, W in (("resnet50", 16, 1, 256, 256), ("resnet50", 17, 5, 224, 160), ("resnet152", 17, 2, 256, 192),
                              ("resnet101", 18, 3, 384, 288), ("resnet50", 17, 130, 128, 96)):
    outs = {}
    for dt in (torch.float32, torch.float16):
        m = pm.deconv(backbone, nc, False)
        m.load_state_dict(synth.fill_pose_state_dict(m.state_dict(), 3))
        m = m.cuda().eval(); m.compute_dtype = dt
        x = synth.pose_crops(3, B, H, W).cuda()
        a = m(x); b = m(x)
        assert torch.equal(a, b) and torch.isfinite(a).all(), (backbone, dt)
        outs[dt] = a
    r = rel(outs[torch.float16], outs[torch.float32])
    print(f"{backbone} K={nc} B={B} {H}x{W}: fp16 vs fp32 rel max err {r:.4f}")
    ok &= r < 0.08
args = types.SimpleNamespace(rgb_max=255.0, fp16=False)
for name, B, H, W in (("FlowNet2S", 1, 64, 64), ("FlowNet2S", 3, 192, 320), ("FlowNet2C", 2, 128, 192), ("FlowNet2CS", 1, 256, 256),
                      ("FlowNet2SD", 2, 128, 128), ("FlowNet2CSS", 1, 128, 192), ("FlowNet2", 1, 192, 256)):
    outs = {}
    for dt in (torch.float32, torch.float16):
        m = getattr(fm, name)(args)
        m.load_state_dict(synth.fill_flow_state_dict(m.state_dict(), 5))
        m = m.cuda().eval(); m.compute_dtype = dt
        x = synth.frame_pairs(5, B, H, W).cuda()
        a = m(x); b = m(x)
        assert torch.equal(a, b) and torch.isfinite(a).all(), (name, dt)
        outs[dt] = a
    epe = torch.norm(outs[torch.float16] - outs[torch.float32], dim=1).mean().item()
    mag = torch.norm(outs[torch.float32], dim=1).mean().item()
    print(f"{name} B={B} {H}x{W}: fp16 EPE {epe:.4f} px at mean |flow| {mag:.2f}")
    ok &= epe < 0.05 * max(mag, 1.0) + 0.1
print("SWEEP", "OK" if ok else "FAILED")
