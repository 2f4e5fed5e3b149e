"""Dev: repeat ft_bottleneck_cluster_fwd on one input and report where runs differ from the first (which images, channel
quarters = phase-3 members, pixel rows) and the workspace's status word.  python tools/dev/bnc_stress.py [N] [runs] [poison]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from flowtrack.pytorch_amd import _lib, synth  # noqa: E402
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, _bottleneck_desc, record_bottleneck  # noqa: E402
from util import make_program, nchw_to_view, run_program, view_to_nchw  # noqa: E402


def bn(seed, name, c):
    return {"weight": synth.uniform(seed, name + "g", (c,), 0.5, 1.5), "bias": synth.normal(seed, name + "b", (c,), 0.1),
            "running_mean": synth.normal(seed, name + "m", (c,), 0.1), "running_var": synth.uniform(seed, name + "v", (c,), 0.5, 1.5), "eps": 1e-5}


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    poison = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    mode = sys.argv[4] if len(sys.argv) > 4 else ""       # "zero": counters zeroed before every run; "quiet": no GPU work between runs
    H, W, P, C = 16, 12, 256, 1024
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 29
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
    ys = ActView(torch.zeros((N, H, W, C), dtype=dtype, device=dev), C, 0)
    ps = make_program()
    record_bottleneck(ps, c1, c2, c3, xv, ys, "s")
    run_program(ps)
    ref = ys.t.clone()
    prog = make_program()
    record_bottleneck(prog, c1, c2, c3, xv, y, "c", cluster=True)
    ws = prog._cluster_ws[(N, H, W)]
    d = _bottleneck_desc(xv, y, P)
    soff = int(_lib.load().ft_bottleneck_cluster_status_offset(ctypes.byref(d)))
    npad = (N + 7) // 8 * 8
    bad_runs = 0
    outs = []
    for k in range(runs):
        y.t.fill_(5.0)
        if poison:
            ws[:soff - 64 * npad].fill_(0x3C)
        if "zero" in mode:
            ws[soff - 64 * npad:soff].zero_()
        torch.cuda.synchronize()
        run_program(prog)
        torch.cuda.synchronize()
        if "quiet" in mode:
            outs.append(y.t.cpu())
            continue
        diff = (y.t.float() - ref.float()).abs() > 0.05
        st = int(ws[soff:soff + 4].view(torch.int32).item())
        if diff.any() or st:
            bad_runs += 1
            idx = diff.nonzero()
            imgs = sorted(set(idx[:, 0].tolist()))
            rows = sorted(set(idx[:, 1].tolist()))
            quarters = sorted(set((idx[:, 3] // 256).tolist()))
            tiles = sorted(set((idx[:, 3] // 32).tolist()))
            print(f"run {k}: status {st} mismatches {int(diff.sum())} images {imgs[:12]}{'...' if len(imgs) > 12 else ''} rows {rows} quarters {quarters} "
                  f"channel tiles {tiles[:16]}{'...' if len(tiles) > 16 else ''}")
            if len(imgs) <= 2:
                for n in imgs:
                    sub = diff[n]
                    print("   image", n, "pixels", sorted(set((sub.nonzero()[:, 0] * W + sub.nonzero()[:, 1]).tolist()))[:40],
                          "got", y.t[n][sub][:4].tolist(), "want", ref[n][sub][:4].tolist())
    if outs:
        rc = ref.cpu().float()
        bad = [int(((o.float() - rc).abs() > 0.05).sum()) for o in outs]
        bad_runs = sum(b > 0 for b in bad)
        print("quiet mode mismatches per run:", bad)
    cnt = ws[soff - 64 * npad:soff].view(torch.int32)[::16]
    print(f"N={N} runs={runs} poison={poison}: {bad_runs} bad runs; counters {sorted(set(cnt[:N].tolist()))}")


if __name__ == "__main__":
    main()
