"""Dev: structure of the elements that differ from the expected constants (FT_CONV_DBG=8192) for deconv3 at batch 64."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import _lib, synth
from flowtrack.pytorch_amd.hip_ops import FusedConv, Program, new_act
N, Cin, H, W, Cout = 64, 256, 16, 12, 256
dev, dtype = torch.device("cuda:0"), torch.float16
lib = _lib.load()
w = synth.normal(35, "w", (Cin, Cout, 4, 4), std=0.05)
layer = FusedConv(w, dtype=dtype, device=dev, stride=2, pad=1, transposed=True, act="relu", label="dbg")
x = new_act(N, H, W, Cin, dtype, dev); x.t[..., :Cin] = synth.normal(35, "x", (N, H, W, Cin)).to(dev, dtype)
y = new_act(N, 2 * H, 2 * W, Cout, dtype, dev)
prog = Program(torch.cuda.Stream()); layer.record(prog, x, y); prog.resolve_choices(); prog._ensure_workspace()
d = prog.conv_records[-1][3]
d.tile_hint = 256 | (256 << 12) | (1 << 24) | (3 << 28)
# expected constant per channel: wave * 16 + i * 4 + rg with wave = wc * 4 + wp: depends on the pixel's row in its tile
for r in range(4):
    y.t.fill_(7.0); torch.cuda.synchronize(); prog.run_eager(); prog.stream.synchronize()
    o = y.t.float().cpu()                         # [N, 32, 24, 256]
    # tile geometry: pixel m = n * 192 + qy * 12 + qx per phase (py, px); tile = m // 256; row = m % 256; wp = row // 64
    n_, oy, ox = torch.meshgrid(torch.arange(N), torch.arange(2 * H), torch.arange(2 * W), indexing="ij")
    m = n_ * (H * W) + (oy // 2) * W + (ox // 2)
    row = m % 256
    wp = row // 64
    ch = torch.arange(256)
    wc, i, rg = ch // 128, (ch % 128) // 32, (ch % 32) // 8
    want = ((wc * 4)[None, None, None, :] + wp[..., None]) * 16 + (i * 4 + rg)[None, None, None, :]
    bad = (o != want.float())
    print(f"run {r}: bad elements {int(bad.sum())}")
    if bad.any():
        idx = bad.nonzero()
        t = (m[idx[:, 0], idx[:, 1], idx[:, 2]] // 256)
        ph = (idx[:, 1] % 2) * 2 + idx[:, 2] % 2
        rr = row[idx[:, 0], idx[:, 1], idx[:, 2]]
        print("   tiles (ptile, phase):", sorted(set(zip(t.tolist(), ph.tolist())))[:24])
        print("   rows in tile // 32 (= wp * 2 + j):", sorted(set((rr // 32).tolist())), " rows % 32:", sorted(set((rr % 32).tolist()))[:40])
        print("   channel chunks:", sorted(set((idx[:, 3] // 8).tolist())))
        vals = o[bad]
        print("   distinct bad values:", sorted(set(vals.tolist()))[:12])
