"""Dev tool: the 8-phase 256 x 256 tile (tile_hint wide level 3, csrc/conv_igemm8.hip) against every other tile variant
the library offers, layer by layer, at BASELINE configs[1] / configs[3] sizes.  hipEvent timing, back-to-back launches.
    python tools/dev/conv8_bench.py [pose|flow|all] [substring,substring...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from flowtrack.pytorch_amd import _lib
from flowtrack.pytorch_amd._lib import check
from flowtrack.pytorch_amd.hip_ops import FusedConv, Program, new_act

# name, N, Cin, H, W, Cout, k, stride, pad, transposed, tail
POSE = [
    ("deconv0_2048", 64, 2048, 8, 6, 256, 4, 2, 1, 1, 0),
    ("deconv3_256", 64, 256, 16, 12, 256, 4, 2, 1, 1, 0),
    ("deconv6_256+heatmap", 64, 256, 32, 24, 256, 4, 2, 1, 1, 17),
    ("deconv6_256", 64, 256, 32, 24, 256, 4, 2, 1, 1, 0),
    ("l3.0.c2_3x3s2_256", 64, 256, 32, 24, 256, 3, 2, 1, 0, 0),
    ("l4.0.c2_3x3s2_512", 64, 512, 16, 12, 512, 3, 2, 1, 0, 0),
    ("l3.c2_3x3_256", 64, 256, 16, 12, 256, 3, 1, 1, 0, 0),
    ("l3.c3_256_1024", 64, 256, 16, 12, 1024, 1, 1, 0, 0, 0),
    ("l3.c1_1024_256", 64, 1024, 16, 12, 256, 1, 1, 0, 0, 0),
    ("l4.c2_3x3_512", 64, 512, 8, 6, 512, 3, 1, 1, 0, 0),
    ("r101_384.deconv6+heatmap_b16", 16, 256, 48, 36, 256, 4, 2, 1, 1, 17),
    ("r101_384.deconv3_b16", 16, 256, 24, 18, 256, 4, 2, 1, 1, 0),
    ("r101_384.deconv0_b16", 16, 2048, 12, 9, 256, 4, 2, 1, 1, 0),
]
FLOW = [
    ("f.conv3_5x5", 16, 128, 96, 128, 256, 5, 2, 2, 0, 0),
    ("f.conv3_1", 16, 256, 48, 64, 256, 3, 1, 1, 0, 0),
    ("f.conv4", 16, 256, 48, 64, 512, 3, 2, 1, 0, 0),
    ("f.conv4_1", 16, 512, 24, 32, 512, 3, 1, 1, 0, 0),
    ("f.conv5", 16, 512, 24, 32, 512, 3, 2, 1, 0, 0),
    ("f.conv5_1", 16, 512, 12, 16, 512, 3, 1, 1, 0, 0),
    ("f.conv6", 16, 512, 12, 16, 1024, 3, 2, 1, 0, 0),
    ("f.conv6_1", 16, 1024, 6, 8, 1024, 3, 1, 1, 0, 0),
    ("f.deconv5", 16, 1024, 6, 8, 512, 4, 2, 1, 1, 0),
]


def fmt(h):
    if h == 0:
        return "heuristic"
    s = f"bp{h & 0xfff} bc{(h >> 12) & 0x1ff} ks{(h >> 24) & 0xf} w{(h >> 28) & 3}"
    if (h >> 30) & 1:
        s += " halo"
    if (h >> 21) & 7:
        s += f" sk{(1, 2, 4, 8, 3, 5, 6, 7)[(h >> 21) & 7]}"
    return s


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "pose"
    cases = {"pose": POSE, "flow": FLOW, "all": POSE + FLOW}[which]
    if len(sys.argv) > 2:
        keys = sys.argv[2].split(",")
        cases = [c for c in cases if any(k in c[0] for k in keys)]
    iters = int(os.environ.get("ITERS", "20"))
    dev, dtype = torch.device("cuda:0"), torch.float16
    lib = _lib.load()
    for (name, N, Cin, H, W, Cout, k, s, p, tr, tail) in cases:
        w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)) * 0.05
        kw = {"tail_weight": torch.randn(tail, Cout, 1, 1) * 0.05, "tail_bias": torch.zeros(tail)} if tail else {}
        layer = FusedConv(w, dtype=dtype, device=dev, stride=s, pad=p, transposed=bool(tr),
                          bn={"weight": torch.ones(Cout), "bias": torch.zeros(Cout), "running_mean": torch.zeros(Cout), "running_var": torch.ones(Cout)},
                          act="relu", label=name, **kw)
        x = new_act(N, H, W, Cin, dtype, dev)
        x.t.normal_()
        if x.cstride > Cin:
            x.t[..., Cin:] = 0
        Ho, Wo = layer.out_hw(H, W)
        y = torch.empty((N, tail, Ho, Wo), dtype=torch.float32, device=dev) if tail else new_act(N, Ho, Wo, Cout, dtype, dev)
        prog = Program(torch.cuda.Stream())
        layer.record(prog, x, y)
        prog.resolve_choices()
        prog._ensure_workspace()
        d = prog.conv_records[-1][3]
        idx = prog.conv_records[-1][1]
        name_, args = prog.calls[idx]
        hints = (ctypes.c_int * 32)()
        n = lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 32)
        sh = prog.stream_handle
        fn = getattr(lib, name_)
        rows = []
        for h in [0] + [int(v) for v in hints[:max(n, 0)]]:
            d.tile_hint = h
            for _ in range(3):
                check(fn(*args, sh))
            e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
            lib.ft_event_create(ctypes.byref(e0)); lib.ft_event_create(ctypes.byref(e1))
            best = 1e9
            for _ in range(3):
                lib.ft_event_record(e0, sh)
                for _ in range(iters):
                    fn(*args, sh)
                lib.ft_event_record(e1, sh); lib.ft_event_synchronize(e1)
                ms = ctypes.c_float(); lib.ft_event_elapsed_ms(e0, e1, ctypes.byref(ms))
                best = min(best, ms.value / iters)
            lib.ft_event_destroy(e0); lib.ft_event_destroy(e1)
            rows.append((best, h))
        fl = prog.flops
        base = rows[0][0]
        best_old = min(r for r in rows if (r[1] >> 28) & 3 != 3)
        best_new = min([r for r in rows if (r[1] >> 28) & 3 == 3], default=None)
        print(f"{name:30s} {fl / 1e9:7.2f} GFLOP  heuristic {base * 1e3:7.1f} us {fl / base / 1e9:7.1f} TF/s | best other {best_old[0] * 1e3:7.1f} us "
              f"{fl / best_old[0] / 1e9:7.1f} TF/s ({fmt(best_old[1])})" +
              (f" | 8-phase {best_new[0] * 1e3:7.1f} us {fl / best_new[0] / 1e9:7.1f} TF/s ({fmt(best_new[1])})" if best_new else " | 8-phase n/a"), flush=True)
        if os.environ.get("ALL"):
            for ms, h in rows:
                print(f"      {fmt(h):32s} {ms * 1e3:7.1f} us {fl / ms / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    main()
