"""Dev: N launches of the 256-plane identity block in one form (cluster | strip) at R50 batch 64, for a PMC pass
(tools/dev/prof_traffic.sh <tag> python tools/dev/bnc_traffic.py <form> 8).  x ping-pongs between two buffers like in the network."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, record_bottleneck
form = sys.argv[1] if len(sys.argv) > 1 else "cluster"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, P, H, W = 64, 256, 16, 12
C = 4 * P
dev, dt = torch.device("cuda:0"), torch.float16
bn = lambda c: {"weight": torch.ones(c), "bias": torch.zeros(c), "running_mean": torch.zeros(c), "running_var": torch.ones(c), "eps": 1e-5}
mk = dict(dtype=dt, device=dev, act="relu")
c1 = FusedConv(synth.normal(1, "w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5), bn=bn(P), **mk)
c2 = FusedConv(synth.normal(1, "w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5), pad=1, bn=bn(P), **mk)
c3 = FusedConv(synth.normal(1, "w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5), bn=bn(C), **mk)
x = ActView(torch.randn((B, H, W, C), device=dev).to(dt), C, 0)
y = ActView(torch.zeros((B, H, W, C), dtype=dt, device=dev), C, 0)
prog = Program(torch.cuda.Stream())
for k in range(n):
    a, b = (x, y) if k % 2 == 0 else (y, x)
    record_bottleneck(prog, c1, c2, c3, a, b, "blk", cluster=(form == "cluster"))
torch.cuda.synchronize()
prog.run_eager(); prog.stream.synchronize()
print(form, "launched", n)
