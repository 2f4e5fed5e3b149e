"""Dev: the in-network correlation at FlowNetC's shape, many launches on one input, alone and with a second stream hammering memory
(the look-ahead ring and the band stores are ordered by hand-counted waits only): every output must equal the first bit for bit and
match the fp32 reference within fp16 tolerance."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
bad = 0
for (B, H, W) in ((16, 48, 64), (3, 47, 61), (8, 24, 32)):
    f = torch.randn((2 * B, H, W, 256), device=dev).half()
    y = torch.zeros((B, H, W, 480), dtype=torch.float16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    def run():
        st = lib.ft_correlation_nhwc_fwd(f[:B].data_ptr(), f[B:].data_ptr(), y.data_ptr(), B, 256, H, W, 20, 2, 256, 480, 32,
                                          _lib.FT_ACT_LEAKY, ctypes.c_float(0.1), _lib.FT_F16, s)
        assert st == 0, st
    run(); torch.cuda.synchronize(); ref = y.clone()
    # fp32 reference on a few displacements
    a, b = f[:B].float(), f[B:].float()
    for (dy, dx) in ((0, 0), (-10, 10), (10, -10), (3, -7)):
        sh = torch.zeros_like(b)
        ys, xs = 2 * dy, 2 * dx
        y0, y1 = max(0, -ys), min(H, H - ys); x0, x1 = max(0, -xs), min(W, W - xs)
        if y1 > y0 and x1 > x0:
            sh[:, y0:y1, x0:x1] = b[:, y0 + ys:y1 + ys, x0 + xs:x1 + xs]
        want = torch.nn.functional.leaky_relu((a * sh).sum(-1) / 256.0, 0.1)
        got = ref[..., 32 + (dy + 10) * 21 + (dx + 10)].float()
        err = (got - want).abs().max().item()
        if err > 2e-2 * max(1.0, want.abs().max().item()):
            bad += 1; print(f"[{B},{H},{W}] displacement ({dy},{dx}): max err {err:.4f}  <<<< BAD")
    side = torch.cuda.Stream()
    junk = torch.empty((256 << 20,), dtype=torch.uint8, device=dev)
    nd = 0
    for it in range(300):
        if it >= 100:
            with torch.cuda.stream(side):
                junk.add_(1)                      # HBM traffic from another stream while the kernel runs
        y.fill_(3.0)
        run()
        torch.cuda.synchronize()
        nd += int(not torch.equal(y[..., 32:473], ref[..., 32:473]))     # (the operator writes its 441 channels only)
    print(f"[{B},{H},{W}]: {nd}/300 launches differ from the first")
    bad += nd
print("CORR STRESS", "OK" if bad == 0 else f"{bad} BAD")
