"""Phase timeline of the streamed fused bottleneck (FT_BNS_DBG=32 stamps s_memtime at phase boundaries of wave 0).
usage: bns_phases.py [P=128|256] [B] ; FT_BNS_VARIANT applies."""
import os, sys
os.environ["FT_BNS_DBG"] = str(32 | int(os.environ.get("FT_BNS_DBG", "0")))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, record_bottleneck
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W = (32, 24) if P == 128 else (16, 12)
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
record_bottleneck(prog, c1, c2, c3, x, y, "a")
torch.cuda.synchronize()
for _ in range(3):
    y.t.zero_()
    prog.run_eager(); prog.stream.synchronize()
var = os.environ.get("FT_BNS_VARIANT", "")
TH = 8 if P == 128 or var == "1" or (var != "2" and B * 2 >= 224) else 4     # rows per strip of the variant bns_plan picks
raw = y.t[:, ::TH, 0, :32].contiguous().view(torch.int64).reshape(-1, 8).cpu()
t = raw.double()                          # the first pixel of every strip carries the stamps
t0 = t[:, 0].min()
names = ["phase1 loop", "T1 epilogue", "phase2 loop", "T2 epilogue", "phase3 quarters", "drain stores"]
d = t[:, 1:7] - t[:, :6]
print(f"P={P} B={B}: {t.shape[0]} strips; kernel span {(t[:, 6].max() - t0):.0f} ticks (100 MHz?); per-strip lifetime mean {(t[:, 6] - t[:, 0]).mean():.0f}")
for i, nme in enumerate(names):
    print(f"  {nme:20s} mean {d[:, i].mean():8.0f}  p10 {d[:, i].quantile(0.1):8.0f}  p90 {d[:, i].quantile(0.9):8.0f}")
if t[:, 7].max() > 0:
    su = t[:, 7] - t[:, 0]
    print(f"  {'start-up (x chunk 0)':20s} mean {su.mean():8.0f}  p10 {su.quantile(0.1):8.0f}  p90 {su.quantile(0.9):8.0f}   (inside phase 1)")
starts = torch.sort(t[:, 0] - t0).values
print("  strip start quantiles:", [int(starts[int(q * (len(starts) - 1))]) for q in (0, 0.25, 0.5, 0.75, 1.0)])
