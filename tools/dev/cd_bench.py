"""Times ft_conv_direct_fwd vs ft_conv2d_fwd on one 1x1 layer, back to back (weights L2-warm) and with a 256 MiB
cache-thrashing copy between launches (weights cold, as inside the network).  usage: cd_bench.py N H W Cin Cout [res]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from flowtrack.pytorch_amd import hip_ops, synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program
N, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
with_res = len(sys.argv) > 6
KK, SS = int(os.environ.get("K", "1")), int(os.environ.get("S", "1"))     # K=3 S=2: the gather form
dev, dt = torch.device("cuda:0"), torch.float16
hip_ops.CONV_DIRECT_MAX_PIXELS = 1 << 22
bn = {"weight": torch.ones(Cout), "bias": torch.zeros(Cout), "running_mean": torch.zeros(Cout), "running_var": torch.ones(Cout), "eps": 1e-5}
conv = FusedConv(synth.normal(1, "w", (Cout, Cin, KK, KK), std=(2.0 / Cin) ** 0.5), stride=SS, pad=KK // 2, bn=bn, act="relu", dtype=dt, device=dev, label="l")
Ho, Wo = conv.out_hw(H, W)
x = ActView(torch.randn((N, H, W, Cin), device=dev).to(dt), Cin, 0)
r = ActView(torch.randn((N, Ho, Wo, Cout), device=dev).to(dt), Cout, 0) if with_res else None
y = ActView(torch.zeros((N, Ho, Wo, Cout), dtype=dt, device=dev), Cout, 0)
big_a = torch.empty(128 << 20, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
fl = 2.0 * N * Ho * Wo * Cin * Cout * KK * KK
for mode in (True, False):
    hip_ops.CONV_DIRECT = mode
    prog = Program(torch.cuda.Stream())
    for _ in range(8):
        conv.record(prog, x, y, residual=r)
        prog.resolve_choices()
    torch.cuda.synchronize()
    prog.run_eager(); prog.stream.synchronize()
    t = prog.time_calls(iters=10)
    warm = sum(ms for _, ms in t) / len(t) * 1e3
    # cold: one launch per iteration, a big copy in between on the same stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    prog1 = Program(torch.cuda.Stream())
    conv.record(prog1, x, y, residual=r)
    prog1.resolve_choices()
    cold = []
    with torch.cuda.stream(prog1.stream):
        for _ in range(6):
            big_b.copy_(big_a)
            ev0.record(prog1.stream)
            prog1.run_eager()
            ev1.record(prog1.stream)
            prog1.stream.synchronize()
            cold.append(ev0.elapsed_time(ev1) * 1e3)
    print(f"{'direct' if mode else 'igemm '} {prog.calls[0][0]:20s} N={N} {H}x{W} {Cin}->{Cout} res={with_res}: warm {warm:6.1f} us ({fl / warm * 1e-6:6.1f} TF/s)   cold {min(cold[1:]):6.1f} us", flush=True)
