"""Dev: does an HBM-bound 1x1 conv overlap with a compute-bound 3x3 conv when both are resident (two streams)?
Decides whether fusing conv2(3x3)+conv3(1x1) into one launch can approach max(t2, t3) instead of t2 + t3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import hip_ops
from flowtrack.pytorch_amd.hip_ops import FusedConv, Program, new_act
hip_ops.benchmark = False
dev = torch.device("cuda:0")
dt = torch.float16
bn = lambda c: {"weight": torch.ones(c), "bias": torch.zeros(c), "running_mean": torch.zeros(c), "running_var": torch.ones(c)}

def mk(N, Cin, H, W, Cout, k, pad, res, stream):
    w = torch.randn(Cout, Cin, k, k) * 0.05
    layer = FusedConv(w, dtype=dt, device=dev, stride=1, pad=pad, bn=bn(Cout), act="relu", label="x")
    x = new_act(N, H, W, Cin, dt, dev); x.t.normal_()
    y = new_act(N, H, W, Cout, dt, dev)
    r = None
    if res:
        r = new_act(N, H, W, Cout, dt, dev); r.t.normal_()
    prog = Program(stream)
    layer.record(prog, x, y, residual=r)
    return prog

def bench(progs, iters=50):
    for p in progs: p.run_eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for p in progs: p.run_eager()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

for name, a, b in (("layer1", (64, 64, 64, 48, 64, 3, 1, False), (64, 64, 64, 48, 256, 1, 0, True)),
                   ("layer2", (64, 128, 32, 24, 128, 3, 1, False), (64, 128, 32, 24, 512, 1, 0, True))):
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    pa, pb = mk(*a, s1), mk(*b, s2)
    pb_same = mk(*b, s1)
    ta, tb = bench([pa]), bench([pb])
    both = bench([pa, pb])
    serial = bench([pa, pb_same])
    print(f"{name}: 3x3 {ta:.1f} us, 1x1+res {tb:.1f} us, same stream {serial:.1f} us, two streams {both:.1f} us")
