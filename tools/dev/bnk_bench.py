"""Times ft_bottleneck_fwd alone (layer1 block at batch B): usage bnk_bench.py [B] [H] [W]; FT_BNK_DBG knobs apply."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, record_bottleneck
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 48
dev, dt = torch.device("cuda:0"), torch.float16
bn = lambda c: {"weight": torch.ones(c), "bias": torch.zeros(c), "running_mean": torch.zeros(c), "running_var": torch.ones(c), "eps": 1e-5}
mk = dict(dtype=dt, device=dev, act="relu")
c1 = FusedConv(synth.normal(1, "w1", (64, 256, 1, 1), std=0.08), bn=bn(64), **mk)
c2 = FusedConv(synth.normal(1, "w2", (64, 64, 3, 3), std=0.06), pad=1, bn=bn(64), **mk)
c3 = FusedConv(synth.normal(1, "w3", (256, 64, 1, 1), std=0.17), bn=bn(256), **mk)
x = ActView(torch.randn((B, H, W, 256), device=dev).to(dt), 256, 0)
y = ActView(torch.zeros((B, H, W, 256), dtype=dt, device=dev), 256, 0)
y2 = ActView(torch.zeros((B, H, W, 256), dtype=dt, device=dev), 256, 0)
prog = Program(torch.cuda.Stream())
for _ in range(4):           # ping-pong like the network does, 4 launches per pass
    record_bottleneck(prog, c1, c2, c3, x, y, "a")
    record_bottleneck(prog, c1, c2, c3, y, y2, "b")
torch.cuda.synchronize()
prog.run_eager(); prog.stream.synchronize()
t = prog.time_calls(iters=10)
us = sum(ms for _, ms in t) / len(t) * 1e3
gb = 2 * B * H * W * 256 * 2 / 1e9
print(f"FT_BNK_DBG={os.environ.get('FT_BNK_DBG', '0'):>3s}  B={B} {H}x{W}: {us:7.1f} us per block   {gb / us * 1e3:6.2f} TB/s algorithmic", flush=True)
