"""Helpers shared by the GPU parity tests: drive single C-ABI launches on torch-allocated buffers."""
import torch

from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, round_up


def nchw_to_view(x_nchw: torch.Tensor, dtype, device, cstride=None, coff=0) -> ActView:
    """NCHW fp32 (CPU) -> ActView on `device` (NHWC, channel stride rounded to 8, padding zero)."""
    N, C, H, W = x_nchw.shape
    cs = cstride or round_up(coff + C, 8)
    buf = torch.zeros((N, H, W, cs), dtype=dtype, device=device)
    buf[..., coff:coff + C] = x_nchw.permute(0, 2, 3, 1).to(device=device, dtype=dtype)
    return ActView(buf, C, coff)


def view_to_nchw(v: ActView) -> torch.Tensor:
    return v.t[..., v.coff:v.coff + v.C].permute(0, 3, 1, 2).float().cpu().contiguous()


def run_program(prog: Program):
    torch.cuda.synchronize()      # buffers were filled on torch's current stream; the program runs on its own
    prog.run_eager()
    prog.stream.synchronize()


def make_program() -> Program:
    return Program(torch.cuda.Stream())
