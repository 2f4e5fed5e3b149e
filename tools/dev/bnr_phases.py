"""Per-section cycle sums of bottleneck_rstat_kernel (FT_BNR_DBG=32: every wave adds up s_memtime deltas per section)."""
import os, sys
os.environ["FT_BNR_DBG"] = str(32 | int(os.environ.get("FT_BNR_DBG", "0")))
os.environ.setdefault("FT_BNK_RSTAT", "2")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, record_bottleneck
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 64, 48
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
nwg = B * 4 if not os.environ.get("FT_BNR_SR") else B * ((H + int(os.environ["FT_BNR_SR"]) - 1) // int(os.environ["FT_BNR_SR"]))
t = y.t.view(torch.int64).reshape(-1)[: nwg * 64].reshape(nwg, 8, 8).cpu().double()
ti = y.t.view(torch.int64).reshape(-1)[: nwg * 64].reshape(nwg, 8, 8).cpu()
st, life = ti[:, :, 5], ti[:, :, 0]
print(f"{nwg} workgroups; first start -> last end {((st + life).max() - st.min()).item()} cycles; start spread {(st.max() - st.min()).item()}; "
      f"end spread {((st + life).max() - (st + life).min()).item()}")
for g, names in ((0, ["conv3 + tile out", "residual issue", "conv1", "barrier"]), (1, ["x issue", "conv2", "wait x", "barrier"])):
    w = t[:, g * 4:(g + 1) * 4, :]
    print(f" G{g}: lifetime mean {w[:, :, 0].mean():.0f}")
    for k, nme in enumerate(names):
        print(f"   {nme:18s} mean {w[:, :, 1 + k].mean():8.0f}   max {w[:, :, 1 + k].max():8.0f}")
