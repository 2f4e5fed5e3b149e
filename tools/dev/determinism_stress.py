"""Dev: every tile variant of a set of layers, run 5x on the same input at a size that recycles workgroups on the CUs;
all runs of a variant must be bit-identical and agree with the heuristic variant (fp16 tolerance)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import synth, hip_ops, _lib
from flowtrack.pytorch_amd.hip_ops import FusedConv, Program, new_act, new_rowpacked_act
hip_ops.benchmark = False
dev = torch.device("cuda:0"); dt = torch.float16
lib = _lib.load()
# name, N, Cin, H, W, Cout, k, stride, pad, transposed, residual
CASES = [("stem", 96, 3, 128, 96, 64, 7, 2, 3, 0, 0), ("flowstem", 24, 6, 192, 256, 64, 7, 2, 3, 0, 0),
         ("3x3_64", 96, 64, 32, 24, 64, 3, 1, 1, 0, 0), ("3x3_256", 64, 256, 16, 12, 256, 3, 1, 1, 0, 0),
         ("1x1_256_64", 96, 256, 32, 24, 64, 1, 1, 0, 0, 0), ("1x1_64_256r", 96, 64, 32, 24, 256, 1, 1, 0, 0, 1),
         ("1x1_1024_256", 64, 1024, 16, 12, 256, 1, 1, 0, 0, 0), ("3x3s2_128", 64, 128, 32, 24, 128, 3, 2, 1, 0, 0),
         ("deconv_256", 64, 256, 16, 12, 256, 4, 2, 1, 1, 0), ("deconv_1026", 16, 1026, 12, 16, 256, 4, 2, 1, 1, 0),
         ("5x5s2_64", 16, 64, 96, 128, 128, 5, 2, 2, 0, 0), ("3x3_512_small", 64, 512, 8, 6, 512, 3, 1, 1, 0, 0)]
bad = 0
for (name, N, Cin, H, W, Cout, k, s, p, tr, res) in CASES:
    w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)) * (2.0 / (Cin * k * k)) ** 0.5
    bn = {"weight": torch.ones(Cout), "bias": torch.zeros(Cout), "running_mean": torch.zeros(Cout), "running_var": torch.ones(Cout)}
    layer = FusedConv(w, dtype=dt, device=dev, stride=s, pad=p, transposed=bool(tr), bn=bn, act="relu", label=name)
    if Cin <= 16:
        x = new_rowpacked_act(N, H, W, Cin, p, dt, dev); x.t[:, :, p:p + W, :Cin].normal_()
    else:
        x = new_act(N, H, W, Cin, dt, dev); x.t[..., :Cin].normal_()
    Ho, Wo = layer.out_hw(H, W)
    y = new_act(N, Ho, Wo, Cout, dt, dev)
    r = None
    if res:
        r = new_act(N, Ho, Wo, Cout, dt, dev); r.t.normal_()
    prog = Program(torch.cuda.Stream())
    layer.record(prog, x, y, residual=r)
    d = prog.conv_records[0][3]
    hints = (ctypes.c_int * 32)(); n = lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 32)
    torch.cuda.synchronize()
    d.tile_hint = 0; prog.run_eager(); prog.stream.synchronize(); ref = y.t.float().clone()
    scale = ref.abs().max().item()
    for h in [0] + [int(v) for v in hints[:n]]:
        d.tile_hint = h
        outs = []
        for i in range(5):
            y.t.fill_(7.0); torch.cuda.synchronize(); prog.run_eager(); prog.stream.synchronize(); outs.append(y.t.clone())
        nondet = sum(int(not torch.equal(outs[0], o)) for o in outs[1:])
        err = (outs[0].float() - ref).abs().max().item()
        flag = "" if (nondet == 0 and err <= 0.02 * max(scale, 1.0)) else "   <<<<<<<< BAD"
        bad += bool(flag)
        if flag or h == 0:
            print(f"{name:16s} hint {h:#010x}: nondeterministic runs {nondet}/4, max |diff| vs heuristic {err:.4f} (scale {scale:.2f}){flag}")
print("STRESS", "OK" if bad == 0 else f"{bad} BAD VARIANTS")
