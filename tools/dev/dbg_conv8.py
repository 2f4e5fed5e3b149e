"""Dev: where do two runs of the 8-phase tile differ?  (race hunting)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flowtrack.pytorch_amd import _lib, synth
from flowtrack.pytorch_amd.hip_ops import FusedConv, Program, new_act

def main():
    N, Cin, H, W, Cout, k, s, p, tr = [int(v) for v in (sys.argv[1:10] if len(sys.argv) > 9 else "64 256 16 12 256 4 2 1 1".split())]
    dev, dtype = torch.device("cuda:0"), torch.float16
    lib = _lib.load()
    w = synth.normal(35, "w", (Cin, Cout, k, k) if tr else (Cout, Cin, k, k), std=0.05)
    layer = FusedConv(w, dtype=dtype, device=dev, stride=s, pad=p, transposed=bool(tr), act="relu", label="dbg")
    x = new_act(N, H, W, Cin, dtype, dev); x.t[..., :Cin] = synth.normal(35, "x", (N, H, W, Cin)).to(dev, dtype)
    Ho, Wo = layer.out_hw(H, W)
    y = new_act(N, Ho, Wo, Cout, dtype, dev)
    prog = Program(torch.cuda.Stream()); layer.record(prog, x, y); prog.resolve_choices(); prog._ensure_workspace()
    d = prog.conv_records[-1][3]
    hints = (ctypes.c_int * 32)(); n = lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 32)
    mine = [int(h) for h in hints[:n] if (int(h) >> 28) & 3 == 3]
    d.tile_hint = 0; torch.cuda.synchronize(); prog.run_eager(); prog.stream.synchronize(); ref = y.t.float().clone()
    for h in mine:
        d.tile_hint = h
        outs = []
        for _ in range(6):
            y.t.fill_(7.0); torch.cuda.synchronize(); prog.run_eager(); prog.stream.synchronize(); outs.append(y.t.float().clone())
        print(f"hint {h:#x} sk {(1, 2, 4, 8, 3, 5, 6, 7)[(h >> 21) & 7]}: max |out - default tile| = {(outs[0] - ref).abs().max().item():.4f}")
        for r, o in enumerate(outs[1:], 1):
            diff = (o != outs[0])
            if diff.any():
                idx = diff.nonzero()
                px = idx[:, 0] * Ho * Wo + idx[:, 1] * Wo + idx[:, 2]
                print(f"  run {r}: {diff.sum().item()} elements differ; pixels {px.unique().numel()}; channels {sorted(set((idx[:, 3] // 8).tolist()))[:40]} (x8);"
                      f" first {idx[0].tolist()} vals {outs[0][tuple(idx[0])].item()} vs {o[tuple(idx[0])].item()} ref {ref[tuple(idx[0])].item()}")
                unwritten = ((o == 7.0) & diff).sum().item()
                print(f"    of which left at the fill value in this run: {unwritten}, in run 0: {((outs[0] == 7.0) & diff).sum().item()}")
            else:
                print(f"  run {r}: identical")
main()
