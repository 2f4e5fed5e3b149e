"""Dev: do the fused bottleneck kernels that stage their input by LDS-DMA behind hand-counted `s_waitcnt vmcnt(N)` survive a
CACHE-COLD input?  (Round 5 found, in the first version of bottleneck_cluster_kernel, that a wave's fully out-of-range LDS-DMA
load retires ahead of older loads and breaks the count; the strip / patch kernels issue such loads for their padding rows.)
Each run: poison y, stream a few hundred MB through torch (evicts x and the weights from L2 / MALL), run the fused block,
compare with the three-launch form.   python tools/dev/lds_dma_oob_stress.py P N H W [runs] [FT_BNS_VARIANT]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
if len(sys.argv) > 6:
    os.environ["FT_BNS_VARIANT"] = sys.argv[6]
from flowtrack.pytorch_amd import synth  # noqa: E402
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, act_stride, record_bottleneck  # noqa: E402
from util import make_program, nchw_to_view, run_program  # noqa: E402


def bn(seed, name, c):
    return {"weight": synth.uniform(seed, name + "g", (c,), 0.5, 1.5), "bias": synth.normal(seed, name + "b", (c,), 0.1),
            "running_mean": synth.normal(seed, name + "m", (c,), 0.1), "running_var": synth.uniform(seed, name + "v", (c,), 0.5, 1.5), "eps": 1e-5}


def main():
    P, N, H, W = (int(v) for v in sys.argv[1:5])
    runs = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    C = 4 * P
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 31
    w1 = synth.normal(seed, "w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, "w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, "w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5)
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn(seed, "1", P), label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn(seed, "2", P), label="conv2", **mk)
    c3 = FusedConv(w3, bn=bn(seed, "3", C), label="conv3", **mk)
    x = synth.normal(seed, "x", (N, C, H, W)).half().float()
    xv = nchw_to_view(x, dtype, dev)
    y = ActView(torch.zeros((N, H, W, C), dtype=dtype, device=dev), C, 0)
    t1v = ActView(torch.zeros((N, H, W, act_stride(P)), dtype=dtype, device=dev), P, 0)
    t2v = ActView(torch.zeros((N, H, W, act_stride(P)), dtype=dtype, device=dev), P, 0)
    y3 = ActView(torch.zeros((N, H, W, C), dtype=dtype, device=dev), C, 0)
    p3 = make_program()
    c1.record(p3, xv, t1v); c2.record(p3, t1v, t2v); c3.record(p3, t2v, y3, residual=xv)
    run_program(p3)
    ref = y3.t.clone()
    prog = make_program()
    record_bottleneck(prog, c1, c2, c3, xv, y, "f")
    junk = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device=dev)
    bad = 0
    for k in range(runs):
        y.t.fill_(5.0)
        junk.add_(1)                       # 512 MB of traffic: nothing of x / the weights is left in the caches
        torch.cuda.synchronize()
        run_program(prog)
        diff = (y.t.float() - ref.float()).abs() > 0.06
        if diff.any():
            bad += 1
            idx = diff.nonzero()
            print(f"run {k}: {int(diff.sum())} mismatches, images {sorted(set(idx[:, 0].tolist()))[:10]} rows {sorted(set(idx[:, 1].tolist()))[:16]}")
    print(f"{prog.calls[0][0]} P={P} N={N} {H}x{W} variant={os.environ.get('FT_BNS_VARIANT', '-')}: {bad} of {runs} cold runs differ from the three launches")


if __name__ == "__main__":
    main()
