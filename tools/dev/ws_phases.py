"""Phase sums of conv5x5s2_wstat_kernel (FT_CD_DBG=32: every wave accumulates s_memtime deltas per phase over its patches).
usage: ws_phases.py [N H W Cout]   (default: FlowNet2S conv2 at 16 x 192 x 256 -> 128)"""
import os, sys
os.environ["FT_CD_DBG"] = os.environ.get("FT_CD_DBG", "32")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import hip_ops, synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program
N, H, W, Cout = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (16, 192, 256, 128)
dev, dt = torch.device("cuda:0"), torch.float16
conv = FusedConv(synth.normal(1, "w", (Cout, 64, 5, 5), std=(2.0 / 1600) ** 0.5), stride=2, pad=2, bias=torch.zeros(Cout), act="leaky", slope=0.1,
                 dtype=dt, device=dev, label="conv2")
x = ActView(torch.randn((N, H, W, 64), device=dev).to(dt), 64, 0)
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
y = ActView(torch.zeros((N, Ho, Wo, Cout), dtype=dt, device=dev), Cout, 0)
prog = Program(torch.cuda.Stream())
conv.record(prog, x, y)
prog.resolve_choices()
assert prog.calls[0][0] == "ft_conv_direct_fwd"
torch.cuda.synchronize()
for _ in range(3):
    prog.run_eager(); prog.stream.synchronize()
t = y.t.view(torch.int64).flatten()[:256 * 8 * 8].reshape(256 * 8, 8).cpu().double()
t = t[t[:, 7] > 0]
names = ["wait for the patch (vmcnt)", "barrier A (+ lgkmcnt 0)", "issue next patch", "MFMA loop + finish of prev", "barrier B", "exchange writes + addresses"]
np_ = t[:, 7].mean()
print(f"{t.shape[0]} waves, {np_:.1f} patches each; loop lifetime mean {t[:, 6].mean():.0f} ticks = {t[:, 6].mean() / np_:.0f} per patch")
for i, nme in enumerate(names):
    print(f"  {nme:30s} per patch: mean {(t[:, i] / t[:, 7]).mean():8.1f}  p10 {(t[:, i] / t[:, 7]).quantile(0.1):8.1f}  p90 {(t[:, i] / t[:, 7]).quantile(0.9):8.1f}")
