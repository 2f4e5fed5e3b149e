import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import FusedConv, new_act, new_rowpacked_act, record_maxpool, record_pack_input
from util import make_program, run_program, view_to_nchw
N, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev, dtype, seed, name = torch.device("cuda:0"), torch.float16, 31, "d"
w = synth.normal(seed, name + ".w", (64, 3, 7, 7), std=(2.0 / 147) ** 0.5)
bn = {"weight": synth.uniform(seed, name + "g", (64,), 0.5, 1.5), "bias": synth.normal(seed, name + "b", (64,), 0.3),
      "running_mean": synth.normal(seed, name + "m", (64,), 0.1), "running_var": synth.uniform(seed, name + "v", (64,), 0.5, 1.5), "eps": 1e-5}
x = synth.normal(seed, name + ".x", (N, 3, H, W)).to(dev)
layer = FusedConv(w, stride=2, pad=3, bn=bn, act="relu", dtype=dtype, device=dev, label=name)
Hp, Wp = H // 4, W // 4
fi = new_rowpacked_act(N, H, W, 3, 5, dtype, dev); pooled = new_act(N, Hp, Wp, 64, dtype, dev)
p1 = make_program(); record_pack_input(p1, x, fi); layer.record(p1, fi, pooled, pool=True); run_program(p1)
si = new_rowpacked_act(N, H, W, 3, 3, dtype, dev); a1 = new_act(N, H // 2, W // 2, 64, dtype, dev); rp = new_act(N, Hp, Wp, 64, dtype, dev)
p2 = make_program(); record_pack_input(p2, x, si); layer.record(p2, si, a1); record_maxpool(p2, a1, rp); run_program(p2)
g, r = view_to_nchw(pooled), view_to_nchw(rp)
d = (g - r).abs()
bad = d > 0
print("mismatch", int(bad.sum()), "of", bad.numel(), "max", float(d.max()))
if bad.any():
    idx = bad.nonzero()
    print("n", idx[:, 0].unique().tolist()[:10], "c", idx[:, 1].unique().tolist()[:20])
    print("y", idx[:, 2].unique().tolist()[:70]); print("x", idx[:, 3].unique().tolist()[:50])
    for i in idx[:8].tolist(): print(i, float(g[tuple(i)]), float(r[tuple(i)]))
