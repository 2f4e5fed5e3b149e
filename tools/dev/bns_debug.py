"""Where does ft_bottleneck_stream_fwd differ from the three launches?  usage: bns_debug.py P N H W [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, act_stride, record_bottleneck
P, N, H, W = (int(v) for v in sys.argv[1:5])
runs = int(sys.argv[5]) if len(sys.argv) > 5 else 3
C = 4 * P
dev, dt = torch.device("cuda:0"), torch.float16
def bn(c, s):
    return {"weight": synth.uniform(s, "g", (c,), 0.5, 1.5), "bias": synth.normal(s, "b", (c,), 0.1),
            "running_mean": synth.normal(s, "m", (c,), 0.1), "running_var": synth.uniform(s, "v", (c,), 0.5, 1.5), "eps": 1e-5}
mk = dict(dtype=dt, device=dev, act="relu")
c1 = FusedConv(synth.normal(1, "w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5), bn=bn(P, 1), **mk)
c2 = FusedConv(synth.normal(1, "w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5), pad=1, bn=bn(P, 2), **mk)
c3 = FusedConv(synth.normal(1, "w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5), bn=bn(C, 3), **mk)
x = ActView(torch.randn((N, H, W, C), device=dev).to(dt), C, 0)
y = ActView(torch.zeros((N, H, W, C), dtype=dt, device=dev), C, 0)
t1 = ActView(torch.zeros((N, H, W, act_stride(P)), dtype=dt, device=dev), P, 0)
t2 = ActView(torch.zeros((N, H, W, act_stride(P)), dtype=dt, device=dev), P, 0)
y3 = ActView(torch.zeros((N, H, W, C), dtype=dt, device=dev), C, 0)
prog = Program(torch.cuda.Stream())
record_bottleneck(prog, c1, c2, c3, x, y, "a")
prog3 = Program(torch.cuda.Stream())
c1.record(prog3, x, t1); c2.record(prog3, t1, t2); c3.record(prog3, t2, y3, residual=x)
torch.cuda.synchronize()
prog3.run_eager(); prog3.stream.synchronize()
ref = y3.t.float()
for r in range(runs):
    y.t.fill_(7.0)
    prog.run_eager(); prog.stream.synchronize()
    d = (y.t.float() - ref).abs()
    bad = d > 0.05 * max(1.0, ref.abs().max().item())
    nb = int(bad.sum())
    print(f"run {r}: max diff {d.max().item():.3f}, {nb} bad of {bad.numel()}")
    if nb:
        idx = bad.nonzero()
        print("  images:", torch.unique(idx[:, 0]).tolist()[:40], "count", len(torch.unique(idx[:, 0])))
        print("  rows:", torch.unique(idx[:, 1]).tolist())
        print("  cols:", torch.unique(idx[:, 2]).tolist())
        ch = torch.unique(idx[:, 3]).tolist()
        print("  channels:", ch[:16], "...", ch[-8:], "count", len(ch))
        n0 = int(idx[0, 0]); sub = idx[idx[:, 0] == n0]
        print(f"  image {n0}: rows {torch.unique(sub[:, 1]).tolist()} cols {torch.unique(sub[:, 2]).tolist()} nch {len(torch.unique(sub[:, 3]))}")
        print("  sample got/ref:", [(round(float(y.t[tuple(i)]), 3), round(float(ref[tuple(i)]), 3)) for i in idx[:6]])
