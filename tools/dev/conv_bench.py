"""Per-layer-shape microbenchmark of ft_conv2d_fwd (dev tool): distinct layer kinds of R50@B=64 and FlowNet2S@B=16."""
import sys, os, ctypes, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from flowtrack.pytorch_amd import _lib
from flowtrack.pytorch_amd._lib import check
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, Program, new_act

# name, N, Cin, H, W, Cout, k, stride, pad, transposed, residual, count-in-net
POSE = [
    ("stem7x7", 64, 3, 256, 192, 64, 7, 2, 3, 0, 0, 1),
    ("l1.c1_64_64", 64, 64, 64, 48, 64, 1, 1, 0, 0, 0, 1),
    ("l1.c2_3x3_64", 64, 64, 64, 48, 64, 3, 1, 1, 0, 0, 3),
    ("l1.c3_64_256r", 64, 64, 64, 48, 256, 1, 1, 0, 0, 1, 3),
    ("l1.c1_256_64", 64, 256, 64, 48, 64, 1, 1, 0, 0, 0, 2),
    ("l2.c1_256_128", 64, 256, 64, 48, 128, 1, 1, 0, 0, 0, 1),
    ("l2.c2_3x3s2_128", 64, 128, 64, 48, 128, 3, 2, 1, 0, 0, 1),
    ("l2.ds_256_512s2", 64, 256, 64, 48, 512, 1, 2, 0, 0, 0, 1),
    ("l2.c3_128_512r", 64, 128, 32, 24, 512, 1, 1, 0, 0, 1, 4),
    ("l2.c1_512_128", 64, 512, 32, 24, 128, 1, 1, 0, 0, 0, 3),
    ("l2.c2_3x3_128", 64, 128, 32, 24, 128, 3, 1, 1, 0, 0, 3),
    ("l3.c1_512_256", 64, 512, 32, 24, 256, 1, 1, 0, 0, 0, 1),
    ("l3.c2_3x3_256", 64, 256, 16, 12, 256, 3, 1, 1, 0, 0, 5),
    ("l3.c3_256_1024r", 64, 256, 16, 12, 1024, 1, 1, 0, 0, 1, 6),
    ("l3.c1_1024_256", 64, 1024, 16, 12, 256, 1, 1, 0, 0, 0, 5),
    ("l4.c1_1024_512", 64, 1024, 16, 12, 512, 1, 1, 0, 0, 0, 1),
    ("l4.c2_3x3_512", 64, 512, 8, 6, 512, 3, 1, 1, 0, 0, 2),
    ("l4.c3_512_2048r", 64, 512, 8, 6, 2048, 1, 1, 0, 0, 1, 3),
    ("l4.c1_2048_512", 64, 2048, 8, 6, 512, 1, 1, 0, 0, 0, 2),
    ("deconv0_2048", 64, 2048, 8, 6, 256, 4, 2, 1, 1, 0, 1),
    ("deconv3_256", 64, 256, 16, 12, 256, 4, 2, 1, 1, 0, 1),
    ("deconv6_256", 64, 256, 32, 24, 256, 4, 2, 1, 1, 0, 1),
    ("heatmap17", 64, 256, 64, 48, 17, 1, 1, 0, 0, 0, 1),
]
FUSION = [
    ("fu.conv0_11", 16, 16, 384, 512, 64, 3, 1, 1, 0, 0, 1),
    ("fu.inter0_82_16", 16, 82, 384, 512, 16, 3, 1, 1, 0, 0, 1),
    ("fu.inter1_162_32", 16, 162, 192, 256, 32, 3, 1, 1, 0, 0, 1),
    ("fu.deconv0_162_16", 16, 162, 192, 256, 16, 4, 2, 1, 1, 0, 1),
    ("fu.deconv1_128_32", 16, 128, 96, 128, 32, 4, 2, 1, 1, 0, 1),
    ("fu.pflow0_16_2", 16, 16, 384, 512, 2, 3, 1, 1, 0, 0, 1),
    ("fu.pflow1_32_2", 16, 32, 192, 256, 2, 3, 1, 1, 0, 0, 1),
]
FLOW = [
    ("f.conv1_7x7", 16, 6, 384, 512, 64, 7, 2, 3, 0, 0, 1),
    ("f.conv2_5x5", 16, 64, 192, 256, 128, 5, 2, 2, 0, 0, 1),
    ("f.conv3_5x5", 16, 128, 96, 128, 256, 5, 2, 2, 0, 0, 1),
    ("f.conv3_1", 16, 256, 48, 64, 256, 3, 1, 1, 0, 0, 1),
    ("f.conv4", 16, 256, 48, 64, 512, 3, 2, 1, 0, 0, 1),
    ("f.conv4_1", 16, 512, 24, 32, 512, 3, 1, 1, 0, 0, 1),
    ("f.conv5_1", 16, 512, 12, 16, 512, 3, 1, 1, 0, 0, 2),
    ("f.conv6_1", 16, 1024, 6, 8, 1024, 3, 1, 1, 0, 0, 1),
    ("f.deconv5", 16, 1024, 6, 8, 512, 4, 2, 1, 1, 0, 1),
    ("f.deconv4_1026", 16, 1026, 12, 16, 256, 4, 2, 1, 1, 0, 1),
    ("f.deconv3_770", 16, 770, 24, 32, 128, 4, 2, 1, 1, 0, 1),
    ("f.deconv2_386", 16, 386, 48, 64, 64, 4, 2, 1, 1, 0, 1),
    ("f.pflow2_194", 16, 194, 96, 128, 2, 3, 1, 1, 0, 0, 1),
]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "pose"
    dtype = torch.float16 if (len(sys.argv) < 3 or sys.argv[2] == "fp16") else torch.float32
    cases = {"pose": POSE, "flow": FLOW, "fusion": FUSION, "all": POSE + FLOW}[which]
    if len(sys.argv) > 3:
        keys = sys.argv[3].split(",")
        cases = [c for c in cases if any(k in c[0] for k in keys)]
    iters = int(os.environ.get("ITERS", "20"))
    dev = torch.device("cuda:0")
    lib = _lib.load()
    tot_ms = tot_fl = 0.0
    esz = 2 if dtype == torch.float16 else 4
    for (name, N, Cin, H, W, Cout, k, s, p, tr, res, cnt) in cases:
        w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k)) * 0.05
        layer = FusedConv(w, dtype=dtype, device=dev, stride=s, pad=p, transposed=bool(tr),
                          bn={"weight": torch.ones(Cout), "bias": torch.zeros(Cout), "running_mean": torch.zeros(Cout), "running_var": torch.ones(Cout)},
                          act="relu", label=name)
        x = new_act(N, H, W, Cin, dtype, dev); x.t.normal_()
        if x.cstride > Cin: x.t[..., Cin:] = 0
        Ho, Wo = layer.out_hw(H, W)
        nchw = Cout % 8 != 0
        y = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=dev) if nchw else new_act(N, Ho, Wo, Cout, dtype, dev)
        r = None
        if res:
            r = new_act(N, Ho, Wo, Cout, dtype, dev); r.t.normal_()
        prog = Program(torch.cuda.Stream())
        layer.record(prog, x, y, residual=r)
        prog.resolve_choices()
        name_, args = prog.calls[0]
        sh = prog.stream_handle
        fn = getattr(lib, name_)
        for _ in range(3):
            check(fn(*args, sh))
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        lib.ft_event_create(ctypes.byref(e0)); lib.ft_event_create(ctypes.byref(e1))
        lib.ft_event_record(e0, sh)
        for _ in range(iters):
            fn(*args, sh)
        lib.ft_event_record(e1, sh); lib.ft_event_synchronize(e1)
        ms = ctypes.c_float(); lib.ft_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        ms = ms.value / iters
        fl = prog.flops
        byt = (N * H * W * Cin + N * Ho * Wo * Cout * (2 if res else 1)) * esz + w.numel() * esz
        print(f"{name:18s} x{cnt} {ms*1e3:8.1f} us {fl/ms/1e9:8.1f} TF/s  min-bytes {byt/1e6:7.1f} MB -> {byt/ms/1e9:6.2f} TB/s")
        tot_ms += ms * cnt; tot_fl += fl * cnt
    print(f"TOTAL (weighted by count) {tot_ms:.3f} ms  {tot_fl/tot_ms/1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
