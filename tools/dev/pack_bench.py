"""ft_pack_nchw_to_nhwc timing: fast 4-pixel path (W % 4 == 0) vs the generic per-pixel path (W = 190)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd.hip_ops import Program, new_rowpacked_act, record_pack_input
for W in (192, 190):
    x = torch.randn((64, 3, 256, W), device="cuda")
    v = new_rowpacked_act(64, 256, W, 3, 5, torch.float16, x.device)
    prog = Program(torch.cuda.Stream())
    for _ in range(4):
        record_pack_input(prog, x, v)
    torch.cuda.synchronize()
    prog.run_eager(); prog.stream.synchronize()
    t = prog.time_calls(iters=10)
    print(f"W={W}: {sum(ms for _, ms in t) / len(t) * 1e3:.1f} us per pack")
