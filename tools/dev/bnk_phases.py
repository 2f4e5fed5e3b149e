"""Phase timeline of the fused bottleneck kernel (FT_BNK_DBG=32 stamps s_memtime at phase boundaries of wave 0)."""
import os, sys
os.environ["FT_BNK_DBG"] = str(32 | int(os.environ.get("FT_BNK_DBG", "0")))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, record_bottleneck
B, H, W = 64, 64, 48
dev, dt = torch.device("cuda:0"), torch.float16
bn = lambda c: {"weight": torch.ones(c), "bias": torch.zeros(c), "running_mean": torch.zeros(c), "running_var": torch.ones(c), "eps": 1e-5}
mk = dict(dtype=dt, device=dev, act="relu")
c1 = FusedConv(synth.normal(1, "w1", (64, 256, 1, 1), std=0.08), bn=bn(64), **mk)
c2 = FusedConv(synth.normal(1, "w2", (64, 64, 3, 3), std=0.06), pad=1, bn=bn(64), **mk)
c3 = FusedConv(synth.normal(1, "w3", (256, 64, 1, 1), std=0.17), bn=bn(256), **mk)
x = ActView(torch.randn((B, H, W, 256), device=dev).to(dt), 256, 0)
y = ActView(torch.zeros((B, H, W, 256), dtype=dt, device=dev), 256, 0)
prog = Program(torch.cuda.Stream())
record_bottleneck(prog, c1, c2, c3, x, y, "a")
torch.cuda.synchronize()
for _ in range(3):
    prog.run_eager(); prog.stream.synchronize()
t = y.t[:, ::8, ::16, :32].contiguous().view(torch.int64).reshape(-1, 8).cpu().double()   # [tiles, 8 stamps]
t0 = t[:, 0].min()
names = ["phase1 loop", "epilogue1", "phase2 loop", "res issue + epilogue2 + barrier", "w3q3 + fb3 + wait all", "phase3 quarters", "drain stores"]
d = t[:, 1:] - t[:, :-1]
print(f"{t.shape[0]} tiles; kernel span {(t[:, 7].max() - t0):.0f} ticks; per-tile lifetime mean {(t[:, 7] - t[:, 0]).mean():.0f} ticks")
for i, nme in enumerate(names):
    print(f"  {nme:34s} mean {d[:, i].mean():8.0f}  p10 {d[:, i].quantile(0.1):8.0f}  p90 {d[:, i].quantile(0.9):8.0f}")
starts = torch.sort(t[:, 0] - t0).values
print("  tile start quantiles:", [int(starts[int(q * (len(starts) - 1))]) for q in (0, 0.25, 0.5, 0.75, 1.0)])
