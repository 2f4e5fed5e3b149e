import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth, hip_ops, _lib
from flowtrack.pytorch_amd.hip_ops import FusedConv, Program, new_act, new_rowpacked_act
hip_ops.benchmark = False
N, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0"); dt = torch.float16
w = synth.normal(1, "w", (64, 3, 7, 7), std=0.1)
bn = {"weight": torch.ones(64), "bias": torch.zeros(64), "running_mean": torch.zeros(64), "running_var": torch.ones(64)}
layer = FusedConv(w, dtype=dt, device=dev, stride=2, pad=3, bn=bn, act="relu", label="stem")
x = new_rowpacked_act(N, H, W, 3, 3, dt, dev)
x.t[:, :, 3:3 + W, :3] = synth.normal(2, "x", (N, H, W, 3)).to(dt).cuda()
y = new_act(N, H // 2, W // 2, 64, dt, dev)
prog = Program(torch.cuda.Stream())
layer.record(prog, x, y)
d = prog.conv_records[0][3]
lib = _lib.load()
hints = (ctypes.c_int * 32)(); n = lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 32)
halo = [h for h in hints[:n] if (h >> 30) & 1]
print("candidates", [hex(h) for h in hints[:n]])
torch.cuda.synchronize()
d.tile_hint = 0; prog.run_eager(); prog.stream.synchronize(); ref = y.t.clone()
d.tile_hint = halo[0]
outs = []
for i in range(6):
    y.t.zero_(); torch.cuda.synchronize()
    prog.run_eager(); prog.stream.synchronize(); outs.append(y.t.clone())
print("halo vs plain max diff", (outs[0].float() - ref.float()).abs().max().item())
for i in range(1, 6):
    dd = (outs[i].float() - outs[0].float()).abs()
    bad = (dd.flatten(1).max(1).values > 0).nonzero().flatten().tolist()
    print("run", i, "max diff vs run 0", dd.max().item(), "images", bad[:10])
    if bad:
        b = bad[0]; idx = (dd[b] > 0).nonzero()
        print("   first diffs (y,x,c):", idx[:6].tolist(), "count", len(idx))
# pattern of wrong tiles vs the plain kernel
dd = (outs[0].float() - ref.float()).abs().amax(dim=3)      # [N, Ho, Wo]
Ho, Wo = dd.shape[1], dd.shape[2]
tiles = dd.unfold(1, 8, 8).unfold(2, 16, 16).amax(dim=(3, 4)) if Ho % 8 == 0 and Wo % 16 == 0 else None
if tiles is not None:
    bad = (tiles > 0.05).nonzero().tolist()
    tpi = tiles.shape[1] * tiles.shape[2]
    print("wrong tiles (n,ty,tx):", bad[:20], "of", tiles.numel())
    print("linear tile ids:", [b[0] * tpi + b[1] * tiles.shape[2] + b[2] for b in bad[:20]])
