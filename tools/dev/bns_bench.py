"""Times ft_bottleneck_stream_fwd alone: usage bns_bench.py [P] [B] [H] [W]; FT_BNS_VARIANT / FT_BNS_DBG apply."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, record_bottleneck
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = int(sys.argv[3]) if len(sys.argv) > 3 else (32 if P == 128 else 16)
W = int(sys.argv[4]) if len(sys.argv) > 4 else (24 if P == 128 else 12)
C = 4 * P
dev, dt = torch.device("cuda:0"), torch.float16
bn = lambda c: {"weight": torch.ones(c), "bias": torch.zeros(c), "running_mean": torch.zeros(c), "running_var": torch.ones(c), "eps": 1e-5}
mk = dict(dtype=dt, device=dev, act="relu")
c1 = FusedConv(synth.normal(1, "w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5), bn=bn(P), **mk)
c2 = FusedConv(synth.normal(1, "w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5), pad=1, bn=bn(P), **mk)
c3 = FusedConv(synth.normal(1, "w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5), bn=bn(C), **mk)
x = ActView(torch.randn((B, H, W, C), device=dev).to(dt), C, 0)
y = ActView(torch.zeros((B, H, W, C), dtype=dt, device=dev), C, 0)
y2 = ActView(torch.zeros((B, H, W, C), dtype=dt, device=dev), C, 0)
prog = Program(torch.cuda.Stream())
for _ in range(4):           # ping-pong like the network does
    record_bottleneck(prog, c1, c2, c3, x, y, "a")
    record_bottleneck(prog, c1, c2, c3, y, y2, "b")
torch.cuda.synchronize()
prog.run_eager(); prog.stream.synchronize()
t = prog.time_calls(iters=10)
us = sum(ms for _, ms in t) / len(t) * 1e3
fl = 2.0 * B * H * W * (C * P + 9 * P * P + P * C)
print(f"FT_BNS_DBG={os.environ.get('FT_BNS_DBG', '0'):>3s} VARIANT={os.environ.get('FT_BNS_VARIANT', '-')}  P={P} B={B} {H}x{W}: {us:7.1f} us per block   {fl / us * 1e-6:7.1f} TFLOP/s", flush=True)
