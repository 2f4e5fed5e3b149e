"""Phase timeline of the cluster bottleneck kernel (FT_BNC_DBG=32 stamps s_memtime at phase boundaries of wave 0 of every member).
usage: bnc_phases.py [B]"""
import os, sys
os.environ["FT_BNC_DBG"] = str(32 | int(os.environ.get("FT_BNC_DBG", "0")))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, record_bottleneck
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P, H, W = 256, 16, 12
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
record_bottleneck(prog, c1, c2, c3, x, y, "a", cluster=True)
torch.cuda.synchronize()
for _ in range(3):
    y.t.zero_()
    prog.run_eager(); prog.stream.synchronize()
flat = y.t.reshape(B, H * W, C)
raw = torch.stack([flat[:, m * 8, :32] for m in range(4)], 1).contiguous().view(torch.int64).reshape(-1, 8).cpu()   # pixel m * 8 of image n: member m
t = raw.double()
t0 = t[:, 0].min()
names = ["phase1 loop (conv1, own rows)", "t1 publish + zero rows + cluster wait", "t1 gather + weight prologue", "phase2 loop (conv2)",
         "K-half reduce + t2 publish + cluster wait", "t2 gather", "phase3 (conv3 + residual + stores issued)"]
d = t[:, 1:8] - t[:, :7]
print(f"B={B}: {t.shape[0]} members; kernel span {(t[:, 7].max() - t0):.0f} ticks (100 MHz); member lifetime mean {(t[:, 7] - t[:, 0]).mean():.0f}")
for i, nme in enumerate(names):
    print(f"  {nme:45s} mean {d[:, i].mean():8.0f}  p10 {d[:, i].quantile(0.1):8.0f}  p90 {d[:, i].quantile(0.9):8.0f}")
starts = torch.sort(t[:, 0] - t0).values
print("  member start quantiles:", [int(starts[int(q * (len(starts) - 1))]) for q in (0, 0.25, 0.5, 0.75, 1.0)])
