"""Phase timeline of ft_conv_direct_fwd (FT_CD_DBG=32 stamps s_memtime in wave 0 of every workgroup).
usage: cd_phases.py N H W Cin Cout [res]"""
import os, sys
os.environ["FT_CD_DBG"] = "32"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import hip_ops, synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program
N, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
with_res = len(sys.argv) > 6
dev, dt = torch.device("cuda:0"), torch.float16
hip_ops.CONV_DIRECT_MAX_PIXELS = 1 << 22
bn = {"weight": torch.ones(Cout), "bias": torch.zeros(Cout), "running_mean": torch.zeros(Cout), "running_var": torch.ones(Cout), "eps": 1e-5}
conv = FusedConv(synth.normal(1, "w", (Cout, Cin, 1, 1), std=(2.0 / Cin) ** 0.5), bn=bn, act="relu", dtype=dt, device=dev, label="l")
x = ActView(torch.randn((N, H, W, Cin), device=dev).to(dt), Cin, 0)
r = ActView(torch.randn((N, H, W, Cout), device=dev).to(dt), Cout, 0) if with_res else None
y = ActView(torch.zeros((N, H, W, Cout), dtype=dt, device=dev), Cout, 0)
prog = Program(torch.cuda.Stream())
conv.record(prog, x, y, residual=r)
prog.resolve_choices()
torch.cuda.synchronize()
for _ in range(3):
    prog.run_eager(); prog.stream.synchronize()
M = N * H * W
bn_ = 256 if prog.conv_records[0][3].Cout % 256 == 0 and (M + 95) // 96 * (Cout // 256) >= 160 else 64
rows = y.t.reshape(M, Cout)[::96]
t = torch.stack([rows[:, c:c + 32].contiguous().view(torch.int64) for c in range(0, Cout, bn_)], 1).reshape(-1, 8).cpu().double()
names = ["prologue -> chunk 0 landed", "chunk walk", "drain + barrier", "epilogue + stores issued", "stores drained"]
d = t[:, 1:6] - t[:, :5]
print(f"{t.shape[0]} workgroups (N-tile {bn_}); lifetime mean {(t[:, 5] - t[:, 0]).mean():.0f} ticks")
for i, nme in enumerate(names):
    print(f"  {nme:28s} mean {d[:, i].mean():8.0f}  p10 {d[:, i].quantile(0.1):8.0f}  p90 {d[:, i].quantile(0.9):8.0f}")
