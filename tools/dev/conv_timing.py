"""Read the per-phase K-loop cycle counters of a FT_CONV_TIMING build (FT_CONV_DBG=32)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "dev"))
import torch
from conv_bench import POSE
from flowtrack.pytorch_amd import _lib
from flowtrack.pytorch_amd.hip_ops import FusedConv, Program, new_act
names = sys.argv[1].split(",")
dev = torch.device("cuda:0")
for (name, N, Cin, H, W, Cout, k, s, p, tr, res, cnt) in POSE:
    if not any(n in name for n in names): continue
    w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)) * 0.05
    layer = FusedConv(w, dtype=torch.float16, device=dev, stride=s, pad=p, transposed=bool(tr), act="relu", label=name)
    x = new_act(N, H, W, Cin, torch.float16, dev); x.t.normal_()
    Ho, Wo = layer.out_hw(H, W)
    y = new_act(N, Ho, Wo, Cout, torch.float16, dev)
    prog = Program(torch.cuda.Stream()); layer.record(prog, x, y); prog.run_eager(); prog.stream.synchronize()
    prog.run_eager(); prog.stream.synchronize()
    raw = y.t.view(torch.int64).flatten()[:24].cpu().tolist()
    for blk, o in (("first", raw[:8]), ("mid", raw[8:])):
        nk = o[7]; tot = o[6]
        lab = ["wait_vmcnt", "barrier", "issue", "lds_wait", "mfma_issue", "loop_tail"]
        print(f"{name:16s} {blk:5s} nk={nk:4d} loop={tot:8d} cyc  per-kstep={tot/max(nk,1):7.1f} | " + " ".join(f"{l}={v/max(nk,1):6.1f}" for l, v in zip(lab, o[:6])))
    for blk, o in (("first", raw[16:19]), ("mid", raw[20:23])):
        print(f"{name:16s} {blk:5s}   issue split per k-step (incl. {2} prologue issues): scalar={o[0]/max(nk,1):6.1f} A_dma={o[1]/max(nk,1):6.1f} B_dma={o[2]/max(nk,1):6.1f}")
