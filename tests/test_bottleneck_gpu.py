"""ft_bottleneck_fwd (whole identity-shortcut Bottleneck in one launch, t1 / t2 in LDS only) vs the CPU oracle for
the stock layers (torch CPU fp32 functional = the reference's own arithmetic for blocks.py:105-120) and vs the three
separate ft_conv2d_fwd launches it replaces."""
import pytest
import torch
import torch.nn.functional as F

from flowtrack.pytorch_amd import hip_ops, synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, act_stride, bottleneck_fusable, record_bottleneck
from util import make_program, nchw_to_view, run_program, view_to_nchw

pytestmark = pytest.mark.gpu

# (name, N, H, W, x channel stride, x channel offset)
CASES = [
    ("r50_map_64x48", 2, 64, 48, 256, 0),            # 8 x 16 patches, exact
    ("r101_map_96x72", 1, 96, 72, 256, 0),           # width divides by 8 only: 16 x 8 patches
    ("ragged_y_40x24", 2, 40, 24, 256, 0),           # 16 x 8 patches, ragged in y
    ("ragged_xy_13x20", 3, 13, 20, 256, 0),          # 8 x 16 patches, ragged both ways
    ("tiny_5x3", 1, 5, 3, 256, 0),                   # one patch, mostly padding
    ("view_offset", 1, 16, 16, 320, 32),             # input is a channel slice of a wider buffer
    ("many_patches_recycle", 24, 64, 48, 256, 0),    # 2304 workgroups: > 4 rounds on 512 slots (LDS reuse across workgroups)
]
# ft_bottleneck_stream_fwd (128 / 256 planes, full-width strips): (name, N, H, W, x channel stride, x channel offset, planes)
STREAM_CASES = [
    ("s128_r50_32x24", 3, 32, 24, 512, 0, 128),      # layer2 of R50 at 256x192: 4 strips of 8 rows, exact
    ("s128_r101_48x36", 1, 48, 36, 512, 0, 128),     # layer2 of R101 at 384x288: strips of 5 rows, last one ragged
    ("s128_ragged_13x20", 2, 13, 20, 512, 0, 128),   # 9-row strips on 13 rows
    ("s128_tiny_5x3", 1, 5, 3, 512, 0, 128),         # one strip, mostly padding lanes
    ("s128_view_offset", 1, 16, 16, 576, 32, 128),   # input is a channel slice of a wider buffer
    ("s128_wide_60", 1, 6, 60, 512, 0, 128),         # width 60: two output rows per strip
    ("s256_r50_16x12", 5, 16, 12, 1024, 0, 256),     # layer3 of R50: few workgroups -> the 4-row strips (64 px on 96)
    ("s256_r101_24x18", 2, 24, 18, 1024, 0, 256),    # layer3 of R101
    ("s256_tiny_5x3", 1, 5, 3, 1024, 0, 256),
    ("s256_view_offset", 1, 9, 7, 1056, 32, 256),
    ("s256_many_recycle", 300, 16, 12, 1024, 0, 256),  # 600 workgroups of 8 rows: > 2 rounds on 256 CUs (LDS reuse across workgroups)
    ("s128_many_recycle", 160, 32, 24, 512, 0, 128),   # 640 workgroups
    ("s256_r101_big_strips", 40, 24, 18, 1024, 0, 256),  # 240 workgroups of 4 rows x 18 (72 px on 108)
    ("s256_small_recycle", 100, 16, 12, 1024, 0, 256),   # 400 workgroups of the 4-row strips: > 1 round on 256 CUs
]


def _bn(seed, name, c):
    return {"weight": synth.uniform(seed, name + "g", (c,), 0.5, 1.5), "bias": synth.normal(seed, name + "b", (c,), 0.1),
            "running_mean": synth.normal(seed, name + "m", (c,), 0.1), "running_var": synth.uniform(seed, name + "v", (c,), 0.5, 1.5),
            "eps": 1e-5}


def _bnf(y, bn):
    return F.batch_norm(y, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], training=False, eps=1e-5)


@pytest.mark.parametrize("case", CASES + STREAM_CASES, ids=[c[0] for c in CASES + STREAM_CASES])
def test_fused_bottleneck_matches_oracle_and_the_three_launches(hip_lib, case):
    name, N, H, W, xcs, xoff = case[:6]
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 21
    P = case[6] if len(case) > 6 else 64
    C = 4 * P
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, name + ".w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5)
    bn1, bn2, bn3 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P), _bn(seed, name + ".bn3", C)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    t1 = F.relu(_bnf(F.conv2d(x, w1), bn1))
    t2 = F.relu(_bnf(F.conv2d(t1, w2, padding=1), bn2))
    want = F.relu(_bnf(F.conv2d(t2, w3), bn3) + x)

    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn2, label="conv2", **mk)
    c3 = FusedConv(w3, bn=bn3, label="conv3", **mk)
    xv = nchw_to_view(x, dtype, dev, cstride=xcs, coff=xoff)
    if xoff:
        xv.t[..., :xoff] = 7.0          # neighbours of the slice must not leak in
    y_fused = ActView(torch.full((N, H, W, C + 32), 3.0, dtype=dtype, device=dev), C, 32)   # output into a slice, too
    assert bottleneck_fusable(c1, c2, c3, xv, y_fused)
    prog = make_program()
    record_bottleneck(prog, c1, c2, c3, xv, y_fused, name)
    run_program(prog)
    got = view_to_nchw(y_fused)
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name}: fused bottleneck vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    assert torch.all(y_fused.t[..., :32] == 3.0), "channels outside the output slice were written"

    # the three launches it replaces: same fp16 roundings of t1 / t2, only the fp32 summation order may differ
    t1v = ActView(torch.zeros((N, H, W, act_stride(P)), dtype=dtype, device=dev), P, 0)
    t2v = ActView(torch.zeros((N, H, W, act_stride(P)), dtype=dtype, device=dev), P, 0)
    y3 = ActView(torch.zeros((N, H, W, C), dtype=dtype, device=dev), C, 0)
    prog3 = make_program()
    c1.record(prog3, xv, t1v)
    c2.record(prog3, t1v, t2v)
    c3.record(prog3, t2v, y3, residual=xv)
    run_program(prog3)
    sep = view_to_nchw(y3)
    diff = (got - sep).abs()
    assert diff.max().item() <= 1e-2 * scale, f"{name}: fused vs separate launches max abs diff {diff.max().item():.3e}"
    if prog.calls[0][0] == "ft_bottleneck_stream_fwd" and hip_ops.FOLD_BOTTLENECK_STREAM:
        # folded operands (round 6): the BatchNorm scale is rounded INTO the fp16 weights instead of multiplying the fp32 sum, so the
        # two paths are one fp16 weight-rounding apart: they agree to a fraction of an fp16 step on average, not bit for bit
        print(f"{name}: folded vs separate launches: {100 * (diff > 0).float().mean().item():.1f} % of outputs differ, mean |diff| {diff.mean().item():.2e}")
        assert diff.mean().item() <= 4e-4 * scale, f"{name}: folded form drifts from the separate launches (mean |diff| {diff.mean().item():.3e})"
    else:
        assert (diff > 0).float().mean().item() < 0.05, "fused and separate launches should agree bit for bit almost everywhere"
    if P != 64:
        assert prog.calls[0][0] == "ft_bottleneck_stream_fwd", prog.calls[0][0]

    # determinism: same bits on a second run (poisoned output first)
    y_fused.t.fill_(5.0)
    run_program(prog)
    assert torch.equal(view_to_nchw(y_fused), got)


S256 = [c for c in STREAM_CASES if c[6] == 256] + [
    ("s256_r101_b16_24x18", 16, 24, 18, 1024, 0, 256),   # configs[2] per-GPU shape: 128 full-width strips -> 256 column halves
    ("s256_odd_width_11x9", 3, 11, 9, 1024, 0, 256),     # 9 columns = 5 + 4: the second half is ragged in x
    ("s256_wide_6x40", 2, 6, 40, 1024, 0, 256),          # 40 columns: four parts of 10
]


@pytest.mark.parametrize("variant", [2, 3], ids=["full_width_strips", "column_split"])
@pytest.mark.parametrize("case", S256, ids=[c[0] for c in S256])
def test_stream_256_variants_match_oracle_and_each_other(hip_lib, case, variant, monkeypatch):
    """Both forms of the 256-plane fused kernel on every 256-plane shape, whatever the library's cost model would pick:
    FT_BNS_VARIANT = 2 (full-width strips of <= 64 pixels) / 3 (column parts with their own x-halo, <= 32 pixels).  Same
    weights, same fp16 roundings of t1 / t2: the two must agree with the oracle and, almost everywhere, with each other."""
    name, N, H, W, xcs, xoff, P = case
    if variant == 2 and (H * W > 64 and W > 32):
        pytest.skip("full-width strips need a row of <= 32 pixels at 256 planes")
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 23
    C = 4 * P
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, name + ".w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5)
    bn1, bn2, bn3 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P), _bn(seed, name + ".bn3", C)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    t1 = F.relu(_bnf(F.conv2d(x, w1), bn1))
    t2 = F.relu(_bnf(F.conv2d(t1, w2, padding=1), bn2))
    want = F.relu(_bnf(F.conv2d(t2, w3), bn3) + x)
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn2, label="conv2", **mk)
    c3 = FusedConv(w3, bn=bn3, label="conv3", **mk)
    xv = nchw_to_view(x, dtype, dev, cstride=xcs, coff=xoff)
    outs = {}
    for v in (variant, 5 - variant):
        if v == 2 and (H * W > 64 and W > 32):
            continue
        monkeypatch.setenv("FT_BNS_VARIANT", str(v))
        y = ActView(torch.full((N, H, W, C + 32), 3.0, dtype=dtype, device=dev), C, 32)
        prog = make_program()
        record_bottleneck(prog, c1, c2, c3, xv, y, name)
        assert prog.calls[0][0] == "ft_bottleneck_stream_fwd"
        run_program(prog)
        got = view_to_nchw(y)
        y.t.fill_(5.0)
        run_program(prog)
        assert torch.equal(view_to_nchw(y), got), f"{name} variant {v}: two runs differ"
        assert torch.all(y.t[..., :32] == 5.0), "channels outside the output slice were written"
        outs[v] = got
    scale = max(1.0, want.abs().max().item())
    err = (outs[variant] - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name} variant {variant}: vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    if len(outs) == 2:
        diff = (outs[2] - outs[3]).abs()
        assert diff.max().item() <= 1e-2 * scale
        assert (diff > 0).float().mean().item() < 0.05, "the two forms should agree bit for bit almost everywhere"


CLUSTER_CASES = [
    ("c256_r50_16x12", 5, 16, 12, 1024, 0, 256),          # layer3 of R50 at 256 x 192: 4 rows per member, six full pixel tiles
    ("c256_r50_b64", 64, 16, 12, 1024, 0, 256),           # the benchmarked batch: 256 workgroups = one per CU
    ("c256_recycle_b100", 100, 16, 12, 1024, 0, 256),     # 416 workgroups: clusters of the second round start as slots free up
    ("c256_view_offset_10x7", 3, 10, 7, 1056, 32, 256),   # ragged rows (3 + 3 + 3 + 1), 70 pixels, input a channel slice
    ("c256_tiny_6x3", 1, 6, 3, 1024, 0, 256),             # two rows per member: member 3 owns none
    ("c256_12x15", 9, 12, 15, 1024, 0, 256),              # 180 pixels: last tile ragged, widest supported row
]


@pytest.mark.parametrize("case", CLUSTER_CASES, ids=[c[0] for c in CLUSTER_CASES])
def test_cluster_256_matches_oracle_and_the_strip_form(hip_lib, case):
    """ft_bottleneck_cluster_fwd (four workgroups per image, t1 / t2 exchanged through global memory inside the launch) vs the
    CPU oracle and vs ft_bottleneck_stream_fwd on the same weight stream: same fp16 roundings of t1 / t2, only the fp32
    summation order of conv2 differs (its K halves meet in LDS).  Five runs with poisoned outputs and exchange buffers in
    between: the hand-off must never read a stale line; the workspace's status word must stay 0 (no spin timed out)."""
    import ctypes
    from flowtrack.pytorch_amd import _lib
    from flowtrack.pytorch_amd.hip_ops import _bottleneck_desc, bottleneck_cluster_supported
    name, N, H, W, xcs, xoff, P = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 29
    C = 4 * P
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, name + ".w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5)
    bn1, bn2, bn3 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P), _bn(seed, name + ".bn3", C)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    t1 = F.relu(_bnf(F.conv2d(x, w1), bn1))
    t2 = F.relu(_bnf(F.conv2d(t1, w2, padding=1), bn2))
    want = F.relu(_bnf(F.conv2d(t2, w3), bn3) + x)
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn2, label="conv2", **mk)
    c3 = FusedConv(w3, bn=bn3, label="conv3", **mk)
    xv = nchw_to_view(x, dtype, dev, cstride=xcs, coff=xoff)
    if xoff:
        xv.t[..., :xoff] = 7.0
    y = ActView(torch.full((N, H, W, C + 32), 3.0, dtype=dtype, device=dev), C, 32)
    assert bottleneck_cluster_supported(xv, y, P)
    prog = make_program()
    record_bottleneck(prog, c1, c2, c3, xv, y, name, cluster=True)
    assert prog.calls[0][0] == "ft_bottleneck_cluster_fwd"
    run_program(prog)
    got = view_to_nchw(y)
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name}: cluster form vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    assert torch.all(y.t[..., :32] == 3.0), "channels outside the output slice were written"
    ws = prog._cluster_ws[(N, H, W)]
    d = _bottleneck_desc(xv, y, P)
    soff = int(_lib.load().ft_bottleneck_cluster_status_offset(ctypes.byref(d)))
    for k in range(5):
        y.t.fill_(5.0)
        ws[:soff - 64 * ((N + 7) // 8 * 8)].fill_(0x3C)       # poison the exchange buffers (not the counters)
        run_program(prog)
        assert torch.equal(view_to_nchw(y), got), f"{name}: run {k + 2} differs from the first"
    assert int(ws[soff:soff + 4].view(torch.int32).item()) == 0, "a cluster hand-off timed out"
    ys = ActView(torch.zeros((N, H, W, C), dtype=dtype, device=dev), C, 0)
    prog_s = make_program()
    record_bottleneck(prog_s, c1, c2, c3, xv, ys, name, fold=False)     # the table form: the very operands the cluster form reads
    assert prog_s.calls[0][0] == "ft_bottleneck_stream_fwd"
    run_program(prog_s)
    diff = (got - view_to_nchw(ys)).abs()
    assert diff.max().item() <= 1e-2 * scale, f"{name}: cluster vs strip form max abs diff {diff.max().item():.3e}"
    assert (diff > 0).float().mean().item() < 0.05, "the two forms should agree bit for bit almost everywhere"


def test_cluster_rejects_other_shapes(hip_lib):
    import ctypes
    from flowtrack.pytorch_amd import _lib
    lib = _lib.load()

    def desc(N=2, H=16, W=12, C=1024, P=256, dtype=torch.float16, **kw):
        d = _lib.BottleneckDesc()
        d.dtype = _lib.dtype_code(dtype)
        d.N, d.H, d.W, d.C, d.P = N, H, W, C, P
        d.x_cstride, d.x_coff, d.y_cstride, d.y_coff = C, 0, C, 0
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    assert lib.ft_bottleneck_cluster_supported(ctypes.byref(desc())) == 0
    assert lib.ft_bottleneck_cluster_workspace_bytes(ctypes.byref(desc())) > 2 * 2 * 192 * 512
    for bad in (desc(H=24, W=18), desc(P=128, C=512), desc(dtype=torch.float32), desc(head_only=1), desc(stride=2), desc(H=16, W=16),
                desc(H=8, W=20), desc(H=9, W=7)):
        assert lib.ft_bottleneck_cluster_supported(ctypes.byref(bad)) != 0
        assert lib.ft_bottleneck_cluster_workspace_bytes(ctypes.byref(bad)) == 0


@pytest.mark.parametrize("variant", [2, 3], ids=["full_width_strips", "column_split"])
@pytest.mark.parametrize("case", S256, ids=[c[0] for c in S256])
def test_stream_256_eight_wave_form_is_bit_identical(hip_lib, case, variant, monkeypatch):
    """FT_BNS_WAVES=8 (round 5): the direct 256-plane kernels with eight waves per workgroup — a wave owns ONE output-channel
    tile, two waves share a SIMD — do the same MFMAs in the same order per (channel tile, pixel tile) as the four-wave form:
    identical bits, on every 256-plane shape and both strip forms; two runs of the eight-wave form agree."""
    name, N, H, W, xcs, xoff, P = case
    if variant == 2 and (H * W > 64 and W > 32):
        pytest.skip("full-width strips need a row of <= 32 pixels at 256 planes")
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 37
    C = 4 * P
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, name + ".w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5)
    bn1, bn2, bn3 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P), _bn(seed, name + ".bn3", C)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    want = F.relu(_bnf(F.conv2d(F.relu(_bnf(F.conv2d(F.relu(_bnf(F.conv2d(x, w1), bn1)), w2, padding=1), bn2)), w3), bn3) + x)
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn2, label="conv2", **mk)
    c3 = FusedConv(w3, bn=bn3, label="conv3", **mk)
    xv = nchw_to_view(x, dtype, dev, cstride=xcs, coff=xoff)
    monkeypatch.setenv("FT_BNS_VARIANT", str(variant))
    outs = {}
    for waves in (4, 8):
        monkeypatch.setenv("FT_BNS_WAVES", str(waves))
        y = ActView(torch.full((N, H, W, C + 32), 3.0, dtype=dtype, device=dev), C, 32)
        prog = make_program()
        record_bottleneck(prog, c1, c2, c3, xv, y, name)
        run_program(prog)
        outs[waves] = view_to_nchw(y)
        y.t.fill_(5.0)
        run_program(prog)
        assert torch.equal(view_to_nchw(y), outs[waves]), f"{name} waves {waves}: two runs differ"
        assert torch.all(y.t[..., :32] == 5.0)
    scale = max(1.0, want.abs().max().item())
    assert (outs[8] - want).abs().max().item() <= 2e-2 * scale
    assert torch.equal(outs[4], outs[8]), f"{name}: eight-wave form differs from the four-wave form ({int((outs[4] != outs[8]).sum())} elements)"


S128 = [c for c in STREAM_CASES if c[6] == 128] + [
    ("s128_r101_b16_48x36", 16, 48, 36, 512, 0, 128),    # configs[2] per-GPU shape: 160 strips of 5 rows -> 256 strips of 3 rows
    ("s128_recycle_small", 40, 48, 36, 512, 0, 128),     # 640 small strips: LDS reuse across workgroups
]


@pytest.mark.parametrize("case", S128, ids=[c[0] for c in S128])
def test_stream_128_strip_sizes_match_oracle_and_each_other(hip_lib, case, monkeypatch):
    """Both strip sizes of the 128-plane fused kernel (layer2's identity blocks, blocks.py:105-120) on every 128-plane shape,
    whatever the cost model would pick: FT_BNS_VARIANT128 = 1 (<= 192 output pixels on <= 256 halo pixels, <128, 4, 3>) / 2 (<= 128
    on <= 192, <128, 3, 2>: the form that fills 256 CUs at ResNet-101 384x288 with 16 crops per GPU).  Same weights, same fp16
    roundings of t1 / t2: both agree with the oracle and, almost everywhere, with each other; without the switch the library picks
    the small strips exactly where they need fewer rounds x work."""
    name, N, H, W, xcs, xoff, P = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 27
    C = 4 * P
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, name + ".w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5)
    bn1, bn2, bn3 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P), _bn(seed, name + ".bn3", C)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    t1 = F.relu(_bnf(F.conv2d(x, w1), bn1))
    t2 = F.relu(_bnf(F.conv2d(t1, w2, padding=1), bn2))
    want = F.relu(_bnf(F.conv2d(t2, w3), bn3) + x)
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn2, label="conv2", **mk)
    c3 = FusedConv(w3, bn=bn3, label="conv3", **mk)
    xv = nchw_to_view(x, dtype, dev, cstride=xcs, coff=xoff)
    outs = {}
    for v in (1, 2, 0):
        if v:
            monkeypatch.setenv("FT_BNS_VARIANT128", str(v))
        else:
            monkeypatch.delenv("FT_BNS_VARIANT128", raising=False)
        y = ActView(torch.full((N, H, W, C + 32), 3.0, dtype=dtype, device=dev), C, 32)
        prog = make_program()
        record_bottleneck(prog, c1, c2, c3, xv, y, name)
        assert prog.calls[0][0] == "ft_bottleneck_stream_fwd"
        run_program(prog)
        got = view_to_nchw(y)
        y.t.fill_(5.0)
        run_program(prog)
        assert torch.equal(view_to_nchw(y), got), f"{name} strips {v}: two runs differ"
        assert torch.all(y.t[..., :32] == 5.0), "channels outside the output slice were written"
        outs[v] = got
    scale = max(1.0, want.abs().max().item())
    for v in (1, 2):
        err = (outs[v] - want).abs().max().item()
        assert err <= 2e-2 * scale, f"{name} strips {v}: vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    diff = (outs[1] - outs[2]).abs()
    assert diff.max().item() <= 1e-2 * scale and (diff > 0).float().mean().item() < 0.05
    assert torch.equal(outs[0], outs[1]) or torch.equal(outs[0], outs[2])
    if name == "s128_r101_b16_48x36":
        assert torch.equal(outs[0], outs[2]) , "160 large strips on 256 CUs: the cost model takes the 256 small ones"


# ft_bottleneck_stream_fwd(head_only, stride 2): conv1 + stride-2 conv2 of the 256-plane entry block (layer3.0): (name, N, H, W, x stride, x offset)
HEAD2_CASES = [("h2_r50_32x24", 3, 32, 24, 512, 0), ("h2_r101_48x36", 2, 48, 36, 512, 0), ("h2_tiny_4x2", 1, 4, 2, 512, 0),
               ("h2_ragged_rows_10x8", 2, 10, 8, 544, 32), ("h2_ragged_last_30x24", 2, 30, 24, 512, 0), ("h2_recycle", 70, 32, 24, 512, 0)]


@pytest.mark.parametrize("case", HEAD2_CASES, ids=[c[0] for c in HEAD2_CASES])
def test_stream_head_stride2_matches_oracle_and_the_two_launches(hip_lib, case):
    from flowtrack.pytorch_amd.hip_ops import bottleneck_head_stream_fusable, record_bottleneck_head_stream
    name, N, H, W, xcs, xoff = case
    dev, dtype, seed, P, C = torch.device("cuda:0"), torch.float16, 27, 256, 512
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    bn1, bn2 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    t1 = F.relu(_bnf(F.conv2d(x, w1), bn1))
    want = F.relu(_bnf(F.conv2d(t1, w2, stride=2, padding=1), bn2))
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, stride=2, pad=1, bn=bn2, label="conv2", **mk)
    xv = nchw_to_view(x, dtype, dev, cstride=xcs, coff=xoff)
    if xoff:
        xv.t[..., :xoff] = 7.0
    Ho, Wo = H // 2, W // 2
    y = ActView(torch.full((N, Ho, Wo, P + 32), 3.0, dtype=dtype, device=dev), P, 32)
    assert bottleneck_head_stream_fusable(c1, c2, xv, y)
    prog = make_program()
    record_bottleneck_head_stream(prog, c1, c2, xv, y, name)
    run_program(prog)
    got = view_to_nchw(y)
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name}: fused head vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    assert torch.all(y.t[..., :32] == 3.0), "channels outside the output slice were written"
    t1v = ActView(torch.zeros((N, H, W, act_stride(P)), dtype=dtype, device=dev), P, 0)
    y2 = ActView(torch.zeros((N, Ho, Wo, P), dtype=dtype, device=dev), P, 0)
    prog2 = make_program()
    c1.record(prog2, xv, t1v)
    c2.record(prog2, t1v, y2)
    run_program(prog2)
    diff = (got - view_to_nchw(y2)).abs()
    assert diff.max().item() <= 1e-2 * scale
    assert (diff > 0).float().mean().item() < 0.05, "fused and separate launches should agree bit for bit almost everywhere"
    y.t.fill_(5.0)
    run_program(prog)
    assert torch.equal(view_to_nchw(y), got)
    # the eight-wave form of the same kernel (FT_BNS_WAVES, read per call): identical bits
    import os
    keep = os.environ.get("FT_BNS_WAVES")
    os.environ["FT_BNS_WAVES"] = "8" if keep != "8" else "4"
    try:
        y.t.fill_(5.0)
        run_program(prog)
        assert torch.equal(view_to_nchw(y), got), "the stride-2 head: eight- and four-wave forms differ"
    finally:
        if keep is None:
            del os.environ["FT_BNS_WAVES"]
        else:
            os.environ["FT_BNS_WAVES"] = keep


HEAD_CASES = [("r50_64x48", 2, 64, 48), ("r101_96x72", 1, 96, 72), ("ragged_13x20", 3, 13, 20), ("recycle", 24, 64, 48)]


@pytest.mark.parametrize("case", HEAD_CASES, ids=[c[0] for c in HEAD_CASES])
def test_fused_bottleneck_head_matches_oracle_and_the_two_launches(hip_lib, case):
    """conv1 + conv2 of the stage's entry block (64 -> 64 -> 64) as one launch (head_only)."""
    from flowtrack.pytorch_amd.hip_ops import bottleneck_head_fusable, record_bottleneck_head
    name, N, H, W = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 23
    C = P = 64
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    bn1, bn2 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    want = F.relu(_bnf(F.conv2d(F.relu(_bnf(F.conv2d(x, w1), bn1)), w2, padding=1), bn2))
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn2, label="conv2", **mk)
    xv = nchw_to_view(x, dtype, dev)
    t2f = ActView(torch.full((N, H, W, P), 3.0, dtype=dtype, device=dev), P, 0)
    assert bottleneck_head_fusable(c1, c2, xv, t2f)
    prog = make_program()
    record_bottleneck_head(prog, c1, c2, xv, t2f, name)
    run_program(prog)
    got = view_to_nchw(t2f)
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name}: fused head vs oracle max abs err {err:.3e}"
    t1v = ActView(torch.zeros((N, H, W, P), dtype=dtype, device=dev), P, 0)
    t2v = ActView(torch.zeros((N, H, W, P), dtype=dtype, device=dev), P, 0)
    prog2 = make_program()
    c1.record(prog2, xv, t1v)
    c2.record(prog2, t1v, t2v)
    run_program(prog2)
    diff = (got - view_to_nchw(t2v)).abs()
    assert diff.max().item() <= 1e-2 * scale and (diff > 0).float().mean().item() < 0.05
    t2f.t.fill_(5.0)
    run_program(prog)
    assert torch.equal(view_to_nchw(t2f), got)


@pytest.mark.parametrize("case", HEAD_CASES + [("view_out", 2, 16, 16)], ids=[c[0] for c in HEAD_CASES] + ["view_out"])
def test_fused_entry_block_matches_oracle_and_the_two_launches(hip_lib, case):
    """The stage's WHOLE entry block (64 -> 64 -> 64 -> 256, projection shortcut K-concatenated with conv3, blocks.py:104-119)
    as one launch (projection = 1) vs the torch-CPU functional form and vs head + FusedShortcutConv."""
    from flowtrack.pytorch_amd.hip_ops import (FusedShortcutConv, bottleneck_entry_fusable, record_bottleneck_entry,
                                               record_bottleneck_head)
    name, N, H, W = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 29
    C = P = 64
    CO = 4 * P
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, name + ".w3", (CO, P, 1, 1), std=(2.0 / P) ** 0.5)
    wd = synth.normal(seed, name + ".wd", (CO, C, 1, 1), std=(2.0 / C) ** 0.5)
    bn1, bn2, bn3, bnd = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P), _bn(seed, name + ".bn3", CO), _bn(seed, name + ".bnd", CO)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    t2 = F.relu(_bnf(F.conv2d(F.relu(_bnf(F.conv2d(x, w1), bn1)), w2, padding=1), bn2))
    want = F.relu(_bnf(F.conv2d(t2, w3), bn3) + _bnf(F.conv2d(x, wd), bnd))
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn2, label="conv2", **mk)
    sc = FusedShortcutConv(w3, bn3, wd, bnd, 1, dtype=dtype, device=dev, act="relu", label="conv3+downsample")
    xv = nchw_to_view(x, dtype, dev)
    off = 32 if name == "view_out" else 0
    yf = ActView(torch.full((N, H, W, CO + off), 3.0, dtype=dtype, device=dev), CO, off)
    assert bottleneck_entry_fusable(c1, c2, sc, xv, yf)
    prog = make_program()
    record_bottleneck_entry(prog, c1, c2, sc, xv, yf, name)
    run_program(prog)
    got = view_to_nchw(yf)
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name}: fused entry block vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    if off:
        assert torch.all(yf.t[..., :off] == 3.0), "channels outside the output slice were written"
    t2v = ActView(torch.zeros((N, H, W, P), dtype=dtype, device=dev), P, 0)
    y2 = ActView(torch.zeros((N, H, W, CO), dtype=dtype, device=dev), CO, 0)
    prog2 = make_program()
    record_bottleneck_head(prog2, c1, c2, xv, t2v, name)
    sc.record(prog2, t2v, xv, y2)
    prog2.resolve_choices()
    run_program(prog2)
    diff = (got - view_to_nchw(y2)).abs()
    assert diff.max().item() <= 1e-2 * scale, f"{name}: fused vs head + shortcut launches max abs diff {diff.max().item():.3e}"
    assert (diff > 0).float().mean().item() < 0.05, "fused and separate launches should agree bit for bit almost everywhere"
    yf.t.fill_(5.0)
    run_program(prog)
    assert torch.equal(view_to_nchw(yf), got)


def test_fused_bottleneck_rejects_other_blocks(hip_lib):
    from flowtrack.pytorch_amd import _lib
    import ctypes
    d = _lib.BottleneckDesc()
    d.dtype, d.N, d.H, d.W, d.C, d.P = _lib.FT_F16, 1, 8, 8, 512, 128
    d.x_cstride = d.y_cstride = 512
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(d)) == 2
    d.C, d.P, d.x_cstride, d.y_cstride, d.dtype = 256, 64, 256, 256, _lib.FT_F32
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(d)) == 2
    d.dtype = _lib.FT_F16
    assert hip_lib.ft_bottleneck_supported(ctypes.byref(d)) == 0


# bottleneck_rstat_kernel (register-stationary strips, csrc/bottleneck_rstat.hip): (name, N, H, W, x channel stride, x channel offset, rows per strip)
RSTAT_CASES = [
    ("rs_r50_b4_16rows", 4, 64, 48, 256, 0, 16),       # the benchmarked strip shape: 16 rows x 48 = 12 steps, halo rows on both sides
    ("rs_r50_b2_whole", 2, 64, 48, 256, 0, 64),        # one strip per image: both halo rows outside the image
    ("rs_ragged_13x20", 3, 13, 20, 256, 0, 5),         # 5 + 5 + 3 rows, 100 pixels per strip: last step ragged
    ("rs_tiny_5x3", 1, 5, 3, 256, 0, 5),               # 15 pixels: one step, every lane on an x border somewhere
    ("rs_view_offset", 1, 16, 16, 320, 32, 7),         # input is a channel slice of a wider buffer
    ("rs_wide_62", 2, 9, 62, 256, 0, 4),               # the widest supported row (T1 ring span 254 of 256)
    ("rs_width_33", 2, 20, 33, 256, 0, 0),             # odd width, the library's own strip height
    ("rs_recycle_b40", 40, 64, 48, 256, 0, 8),         # 320 workgroups: a second round on 256 CUs (LDS reuse across workgroups)
]


@pytest.mark.parametrize("case", RSTAT_CASES, ids=[c[0] for c in RSTAT_CASES])
def test_register_stationary_strip_form_matches_oracle_and_the_patch_form(hip_lib, case, monkeypatch):
    """FT_BNK_RSTAT = 2 runs bottleneck_rstat_kernel wherever the shape fits (FT_BNR_SR = rows per strip), 0 the patch kernel.
    The strip form folds each BatchNorm scale into its fp16 weights (one rounding of w * scale instead of one of w) and adds the
    shift and the residual inside the matrix product: the same mathematics with differently placed roundings, so it is held to the
    oracle's tolerance and to fp16-rounding distance from the patch form, not to its bits."""
    name, N, H, W, xcs, xoff, sr = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 29
    P, C = 64, 256
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, name + ".w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5)
    bn1, bn2, bn3 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P), _bn(seed, name + ".bn3", C)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    t1 = F.relu(_bnf(F.conv2d(x, w1), bn1))
    t2 = F.relu(_bnf(F.conv2d(t1, w2, padding=1), bn2))
    want = F.relu(_bnf(F.conv2d(t2, w3), bn3) + x)
    mk = dict(dtype=dtype, device=dev, act="relu")
    c1 = FusedConv(w1, bn=bn1, label="conv1", **mk)
    c2 = FusedConv(w2, pad=1, bn=bn2, label="conv2", **mk)
    c3 = FusedConv(w3, bn=bn3, label="conv3", **mk)
    xv = nchw_to_view(x, dtype, dev, cstride=xcs, coff=xoff)
    if xoff:
        xv.t[..., :xoff] = 7.0
    outs = {}
    for mode in (2, 0):
        monkeypatch.setenv("FT_BNK_RSTAT", str(mode))
        if sr:
            monkeypatch.setenv("FT_BNR_SR", str(sr))
        y = ActView(torch.full((N, H, W, C + 32), 3.0, dtype=dtype, device=dev), C, 32)
        prog = make_program()
        record_bottleneck(prog, c1, c2, c3, xv, y, name)
        assert prog.calls[0][0] == ("ft_bottleneck_rstat_fwd" if mode else "ft_bottleneck_fwd")
        run_program(prog)
        got = view_to_nchw(y)
        assert torch.all(y.t[..., :32] == 3.0), "channels outside the output slice were written"
        for _ in range(3):                       # same bits on every run (poisoned output first)
            y.t.fill_(5.0)
            run_program(prog)
            assert torch.equal(view_to_nchw(y), got), f"{name} mode {mode}: two runs differ"
        outs[mode] = got
    scale = max(1.0, want.abs().max().item())
    err = {m: (outs[m] - want).abs().max().item() for m in outs}
    assert err[2] <= 2e-2 * scale, f"{name}: strip form vs oracle max abs err {err[2]:.3e} (scale {scale:.2f})"
    assert err[2] <= 1.5 * err[0] + 1e-3 * scale, f"{name}: strip form {err[2]:.3e} vs patch form {err[0]:.3e} from the oracle"
    diff = (outs[2] - outs[0]).abs()
    assert diff.max().item() <= 1e-2 * scale, f"{name}: strip form vs patch form max abs diff {diff.max().item():.3e}"
    assert diff.mean().item() <= 5e-4 * scale, f"{name}: strip form vs patch form mean abs diff {diff.mean().item():.3e}"


@pytest.mark.parametrize("planes", [128, 256])
def test_stream_folded_form_edge_cases(hip_lib, planes):
    """The folded operands (round 6) where they could go wrong: NEGATIVE BatchNorm gammas (the folded weights change sign), shifts of
    a few hundred (the (hi, lo) fp16 pair must carry them: hi alone is 0.25 off at 300), a zero gamma (a channel that is its shift),
    and a shift beyond the fp16 range — there hip_ops must fall back to the table form (and still match the oracle)."""
    P, C = planes, 4 * planes
    N, H, W = (2, 16, 12) if P == 256 else (2, 32, 24)
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 41
    name = f"fold_edge_{P}"
    w1 = synth.normal(seed, name + ".w1", (P, C, 1, 1), std=(2.0 / C) ** 0.5)
    w2 = synth.normal(seed, name + ".w2", (P, P, 3, 3), std=(2.0 / (9 * P)) ** 0.5)
    w3 = synth.normal(seed, name + ".w3", (C, P, 1, 1), std=(2.0 / P) ** 0.5)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    for case in ("negative_gamma_big_shift", "shift_beyond_fp16"):
        bn1, bn2, bn3 = _bn(seed, name + ".bn1", P), _bn(seed, name + ".bn2", P), _bn(seed, name + ".bn3", C)
        sign = torch.where(synth.uniform(seed, name + ".sg", (P,)) < 0.4, -1.0, 1.0)
        bn1["weight"] = bn1["weight"] * sign
        bn2["weight"] = bn2["weight"] * torch.flip(sign, dims=[0])
        bn2["weight"][5] = 0.0
        bn3["weight"] = bn3["weight"] * torch.where(synth.uniform(seed, name + ".sg3", (C,)) < 0.5, -1.0, 1.0)
        bn3["bias"] = bn3["bias"] + synth.normal(seed, name + ".big", (C,), std=150.0)          # shifts of a few hundred on the output
        bn1["bias"][3] = 300.3
        if case == "shift_beyond_fp16":
            bn3["bias"][7] = 1.0e5
        t1 = F.relu(_bnf(F.conv2d(x, w1), bn1))
        t2 = F.relu(_bnf(F.conv2d(t1, w2, padding=1), bn2))
        want = F.relu(_bnf(F.conv2d(t2, w3), bn3) + x)
        mk = dict(dtype=dtype, device=dev, act="relu")
        c1, c2, c3 = FusedConv(w1, bn=bn1, **mk), FusedConv(w2, pad=1, bn=bn2, **mk), FusedConv(w3, bn=bn3, **mk)
        xv = nchw_to_view(x, dtype, dev)
        y = ActView(torch.zeros((N, H, W, C), dtype=dtype, device=dev), C, 0)
        prog = make_program()
        record_bottleneck(prog, c1, c2, c3, xv, y, name)
        assert prog.calls[0][0] == "ft_bottleneck_stream_fwd"
        folded = bool(prog.calls[0][1][0]._obj.folded)
        assert folded == (case != "shift_beyond_fp16"), f"{case}: folded = {folded}"
        run_program(prog)
        got = view_to_nchw(y)
        fin = torch.isfinite(want) & (want.abs() < 6.0e4)                  # (the 1e5 channel overflows fp16 on every path)
        # t1 carries a channel of ~300 into conv2 and the outputs reach several hundred: the yardstick is the TABLE form on the same
        # operands (fp32 scale / shift, the same fp16 storage of t1 / t2 / y): the folded form must be as close to the oracle as it is
        yt = ActView(torch.zeros((N, H, W, C), dtype=dtype, device=dev), C, 0)
        prog_t = make_program()
        record_bottleneck(prog_t, c1, c2, c3, xv, yt, name, fold=False)
        run_program(prog_t)
        rel = lambda a: ((a - want).abs() / (1.0 + want.abs()))[fin]
        err, err_t = rel(got).max().item(), rel(view_to_nchw(yt)).max().item()
        print(f"{name} {case}: max relative err folded {err:.3e} / table form {err_t:.3e}; mean {rel(got).mean().item():.2e} / {rel(view_to_nchw(yt)).mean().item():.2e}")
        assert err <= 1.5 * err_t + 1e-3 and rel(got).mean().item() <= 1.5 * rel(view_to_nchw(yt)).mean().item() + 1e-5, f"{case}: folded {err:.3e} vs table {err_t:.3e}"
        if case == "shift_beyond_fp16":
            assert torch.isinf(got[:, 7]).all() and (got[:, 7] > 0).all()
