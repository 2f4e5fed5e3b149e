import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth, hip_ops
from flowtrack.pytorch_amd.pose import models as pm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 130
m = pm.deconv("resnet50", 17, False)
m.load_state_dict(synth.fill_pose_state_dict(m.state_dict(), 3))
m = m.cuda().eval(); m.compute_dtype = torch.float16
x = synth.pose_crops(3, B, 128, 96).cuda()
a = m(x); b = m(x); c = m(x)
print("finite", torch.isfinite(a).all().item(), torch.isfinite(b).all().item(), "a==b", torch.equal(a, b), "b==c", torch.equal(b, c),
      "max|a-b|", (a - b).abs().max().item(), "max|b-c|", (b - c).abs().max().item(), "range", a.min().item(), a.max().item())
d = (a - b).abs().flatten(1).max(1).values
print("rows differing:", (d > 0).nonzero().flatten().tolist()[:20])
