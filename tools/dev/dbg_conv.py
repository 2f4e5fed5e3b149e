import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch, torch.nn.functional as F
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv
from util import make_program, nchw_to_view, run_program, view_to_nchw
dev = torch.device("cuda:0")
for dtype in (torch.float32, torch.float16):
    for (Cin, Cout, k) in ((64, 256, 1), (8, 32, 1), (16, 32, 1)):
        N, H, W = 1, 8, 16
        torch.manual_seed(0)
        x = torch.randn(N, Cin, H, W).half().float(); w = torch.randn(Cout, Cin, k, k).half().float() * 0.1
        layer = FusedConv(w, dtype=dtype, device=dev, stride=1, pad=k // 2, label="dbg")
        xv = nchw_to_view(x, dtype, dev)
        y = torch.empty((N, Cout, H, W), dtype=torch.float32, device=dev)
        prog = make_program(); layer.record(prog, xv, y); run_program(prog)
        want = F.conv2d(x, w, padding=k // 2)
        d = (y.cpu() - want)
        print(dtype, Cin, Cout, k, "maxerr", d.abs().max().item(), "packed", tuple(layer.w.shape))
        if d.abs().max() > 1e-2:
            print(" got ", y.cpu()[0, :4, 0, :6]); print(" want", want[0, :4, 0, :6])
            # which k contributions are present? use one-hot x
            for ci in range(min(Cin, 8)):
                xo = torch.zeros(N, Cin, H, W); xo[:, ci] = 1.0
                xv2 = nchw_to_view(xo, dtype, dev); prog = make_program(); layer.record(prog, xv2, y); run_program(prog)
                wnt = F.conv2d(xo, w)
                print("  ci", ci, "got", y.cpu()[0, :3, 0, 0].tolist(), "want", wnt[0, :3, 0, 0].tolist())
