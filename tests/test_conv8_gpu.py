"""The 256 x 256 8-phase implicit-GEMM tile (csrc/conv_igemm8.hip, tile_hint wide level 3) vs the CPU oracle for the stock
layers (torch CPU fp32 functional: the arithmetic the reference runs for nn.Conv2d / ConvTranspose2d / BatchNorm2d,
lib/pose/models/pose_deconv.py:19-30, lib/pose/models/blocks.py:95-97, lib/flownet/networks/FlowNetS.py:24-45), through
the C ABI.  The schedule keeps LDS-DMA in flight across barriers and refills half-tiles while their K-tile is still being
multiplied, so besides parity every case is a race screen: repeated runs must be bit-identical, at sizes where each CU
recycles workgroups."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, act_stride, new_act
from test_conv_gpu import _reference
from util import make_program, nchw_to_view, run_program, view_to_nchw

pytestmark = pytest.mark.gpu

WIDE8 = 3


def hints8(hip_lib, d):
    hints = (ctypes.c_int * 64)()
    n = hip_lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 64)
    assert n >= 0
    return [int(h) for h in hints[:n] if (int(h) >> 28) & 3 == WIDE8], [int(h) for h in hints[:n]]


CASES8 = [
    # (name, N, Cin, H, W, Cout, k, stride, pad, transposed, bias, bn, act, residual, wants_splitk)
    ("deconv_256_ragged", 5, 256, 13, 9, 256, 4, 2, 1, True, False, True, "relu", False, False),
    ("deconv_2048_k8192", 2, 2048, 8, 6, 256, 4, 2, 1, True, False, True, "relu", False, True),
    ("3x3_s2_128_256", 3, 128, 31, 23, 256, 3, 2, 1, False, False, True, "relu", False, True),
    ("3x3_s1_cin64_odd_ktiles", 2, 64, 16, 12, 256, 3, 1, 1, False, False, True, "relu", False, False),
    ("1x1_512_512_two_ctiles", 3, 512, 16, 12, 512, 1, 1, 0, False, False, True, "relu", False, False),
    ("1x1_128_two_ktiles", 2, 128, 17, 13, 256, 1, 1, 0, False, True, False, "leaky", False, False),
    ("5x5_s2_128_256_bias_leaky", 2, 128, 24, 32, 256, 5, 2, 2, False, True, False, "leaky", False, False),
    ("3x3_cout1024_k4608", 1, 512, 6, 8, 1024, 3, 2, 1, False, True, False, "leaky", False, True),
    ("3x3_coff_views", 2, 256, 12, 16, 256, 3, 1, 1, False, False, True, "relu", False, False),
]


@pytest.mark.parametrize("case", CASES8, ids=[c[0] for c in CASES8])
def test_igemm8_matches_oracle(hip_lib, case):
    name, N, Cin, H, W, Cout, k, stride, pad, transposed, use_bias, use_bn, act, has_res, wants_sk = case
    dev, dtype = torch.device("cuda:0"), torch.float16
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    fan = (Cin * 4) if transposed else Cin * k * k
    w = synth.normal(31, name + ".w", wshape, std=(2.0 / fan) ** 0.5).half().float()
    x = synth.normal(31, name + ".x", (N, Cin, H, W)).half().float()
    bias = synth.normal(31, name + ".b", (Cout,), 0.1) if use_bias else None
    bn = None
    if use_bn:
        bn = {"weight": synth.uniform(31, name + ".g", (Cout,), 0.5, 1.5), "bias": synth.normal(31, name + ".be", (Cout,), 0.1),
              "running_mean": synth.normal(31, name + ".m", (Cout,), 0.1), "running_var": synth.uniform(31, name + ".v", (Cout,), 0.5, 1.5),
              "eps": 1e-5}
    layer = FusedConv(w, dtype=dtype, device=dev, stride=stride, pad=pad, transposed=transposed, bias=bias, bn=bn, act=act, slope=0.1,
                      label=name)
    Ho, Wo = layer.out_hw(H, W)
    res = synth.normal(31, name + ".r", (N, Cout, Ho, Wo)).half().float() if has_res else None
    want = _reference(x, w, bias, bn, stride, pad, transposed, act, res)
    if "coff" in name:      # channel windows inside wider buffers on both sides (concat slices)
        xv = nchw_to_view(x, dtype, dev, cstride=Cin + 64, coff=32)
        ybuf = torch.full((N, Ho, Wo, Cout + 64), 3.0, dtype=dtype, device=dev)
        yv = ActView(ybuf, Cout, 24)
    else:
        xv = nchw_to_view(x, dtype, dev, cstride=act_stride(Cin))
        yv = ActView(torch.zeros((N, Ho, Wo, act_stride(Cout)), dtype=dtype, device=dev), Cout, 0)
    prog = make_program()
    layer.record(prog, xv, yv, residual=nchw_to_view(res, dtype, dev) if res is not None else None)
    d = prog.conv_records[-1][3]
    mine, _ = hints8(hip_lib, d)
    assert mine, f"{name}: the 8-phase tile is not offered"
    if wants_sk:
        assert any((h >> 21) & 7 for h in mine), f"{name}: no split-K form of the 8-phase tile offered"
        # the odd K splits (codes 4..7 = 3, 5, 6, 7 slices: offered only where they fill one round of CUs better than a power of
        # two, e.g. deconv.0 at batch 64 = 48 tiles x 5) forced onto these small cases as well
        base = next(h for h in mine if not (h >> 21) & 7)
        mine = mine + [base | (code << 21) for code in (4, 5, 6, 7)]
    scale = max(1.0, want.abs().max().item())
    for h in mine:
        d.tile_hint = h
        outs = []
        for _ in range(3):
            yv.t[..., yv.coff:yv.coff + Cout].fill_(9.0)
            run_program(prog)
            outs.append(yv.t.clone())
        assert all(torch.equal(outs[0], o) for o in outs[1:]), f"{name} hint {h:#x}: runs differ"
        err = (view_to_nchw(yv) - want).abs().max().item()
        assert err <= 2e-2 * scale, f"{name} hint {h:#x} (split-K code {(h >> 21) & 7}): max abs err {err:.3e} (scale {scale:.2f})"
    if "coff" in name:
        assert torch.all(yv.t[..., :24] == 3.0) and torch.all(yv.t[..., 24 + Cout:] == 3.0), "wrote outside its channel slice"


TAIL8 = [("deconv256_heatmap17", 3, 256, 13, 10, 256, 4, 2, 1, True, 17), ("conv1x1_256_tail1", 1, 512, 12, 10, 256, 1, 1, 0, False, 1),
         ("conv3x3_256_tail24", 2, 128, 20, 9, 256, 3, 1, 1, False, 24)]


@pytest.mark.parametrize("nchw", [True, False], ids=["nchw_f32", "nhwc_f16"])
@pytest.mark.parametrize("case", TAIL8, ids=[c[0] for c in TAIL8])
def test_igemm8_fused_tail_matches_oracle(hip_lib, case, nchw):
    """Wt . relu(bn(conv(x))) + bt (pose_deconv.py:43-45) on the 8-phase tile, straight from the accumulator registers (up to 24
    tail outputs: the two channel halves of a workgroup exchange 12 partial-sum registers per lane)."""
    name, N, Cin, H, W, Cout, k, stride, pad, transposed, nt = case
    dev, dtype = torch.device("cuda:0"), torch.float16
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    fan = (Cin * 4) if transposed else Cin * k * k
    w = synth.normal(33, name + ".w", wshape, std=(2.0 / fan) ** 0.5).half().float()
    x = synth.normal(33, name + ".x", (N, Cin, H, W)).half().float()
    bn = {"weight": synth.uniform(33, name + ".g", (Cout,), 0.5, 1.5), "bias": synth.normal(33, name + ".be", (Cout,), 0.1),
          "running_mean": synth.normal(33, name + ".m", (Cout,), 0.1), "running_var": synth.uniform(33, name + ".v", (Cout,), 0.5, 1.5),
          "eps": 1e-5}
    wt = synth.normal(33, name + ".wt", (nt, Cout, 1, 1), std=(1.0 / Cout) ** 0.5).half().float()
    bt = synth.normal(33, name + ".bt", (nt,), 0.2)
    mid = _reference(x, w, None, bn, stride, pad, transposed, "relu", None).half().float()
    want = F.conv2d(mid, wt, bt)
    layer = FusedConv(w, dtype=dtype, device=dev, stride=stride, pad=pad, transposed=transposed, bn=bn, act="relu", label=name,
                      tail_weight=wt, tail_bias=bt)
    Ho, Wo = layer.out_hw(H, W)
    xv = nchw_to_view(x, dtype, dev, cstride=act_stride(Cin))
    prog = make_program()
    if nchw:
        y = torch.full((N, nt, Ho, Wo), 5.0, dtype=torch.float32, device=dev)
        layer.record(prog, xv, y)
    else:
        ybuf = torch.full((N, Ho, Wo, 40), 7.0, dtype=dtype, device=dev)
        yv = ActView(ybuf, nt, 4)
        layer.record(prog, xv, yv)
    d = prog.conv_records[-1][3]
    mine, offered = hints8(hip_lib, d)
    assert len(mine) == 1 and len(offered) == 2, "a 256-channel tail layer offers the 128-pixel default and the 8-phase tile"
    scale = max(1.0, want.abs().max().item())
    got = {}
    for h in offered:
        d.tile_hint = h
        run_program(prog)
        got[h] = y.cpu().clone() if nchw else view_to_nchw(yv)
        err = (got[h] - want).abs().max().item()
        assert err <= (2e-3 if nchw else 1e-2) * scale, f"{name} hint {h:#x}: max abs err {err:.3e} (scale {scale:.2f})"
    if not nchw:
        assert torch.all(ybuf[..., :4] == 7.0) and torch.all(ybuf[..., 4 + nt:] == 7.0), "wrote outside its channel slice"


FULL8 = [
    # the layers the tile is meant for, at BASELINE configs[1] / configs[3] sizes: (name, N, Cin, H, W, Cout, k, s, p, transposed, tail)
    ("pose_deconv6_heatmap", 64, 256, 32, 24, 256, 4, 2, 1, True, 17),
    ("pose_deconv3", 64, 256, 16, 12, 256, 4, 2, 1, True, 0),
    ("pose_deconv0", 64, 2048, 8, 6, 256, 4, 2, 1, True, 0),
    ("pose_layer3_0_conv2_s2", 64, 256, 32, 24, 256, 3, 2, 1, False, 0),
    ("flow_conv3_1", 16, 256, 48, 64, 256, 3, 1, 1, False, 0),
]


@pytest.mark.parametrize("case", FULL8, ids=[c[0] for c in FULL8])
def test_igemm8_full_size_is_deterministic_and_equals_the_default_tile(hip_lib, case):
    """Hundreds of workgroups per launch, several per CU one after the other: four runs bit-identical, and within fp16
    rounding of the library's default tile for the same layer (both accumulate in fp32; only the summation order differs)."""
    name, N, Cin, H, W, Cout, k, stride, pad, transposed, nt = case
    dev, dtype = torch.device("cuda:0"), torch.float16
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    fan = (Cin * 4) if transposed else Cin * k * k
    w = synth.normal(35, name + ".w", wshape, std=(2.0 / fan) ** 0.5)
    bn = {"weight": synth.uniform(35, name + ".g", (Cout,), 0.5, 1.5), "bias": synth.normal(35, name + ".be", (Cout,), 0.1),
          "running_mean": synth.normal(35, name + ".m", (Cout,), 0.1), "running_var": synth.uniform(35, name + ".v", (Cout,), 0.5, 1.5),
          "eps": 1e-5}
    kw = {}
    if nt:
        kw = {"tail_weight": synth.normal(35, name + ".wt", (nt, Cout, 1, 1), std=(1.0 / Cout) ** 0.5), "tail_bias": synth.normal(35, name + ".bt", (nt,), 0.2)}
    layer = FusedConv(w, dtype=dtype, device=dev, stride=stride, pad=pad, transposed=transposed, bn=bn, act="relu", label=name, **kw)
    xv = new_act(N, H, W, Cin, dtype, dev)
    xv.t[..., :Cin] = synth.normal(35, name + ".x", (N, H, W, Cin)).to(device=dev, dtype=dtype)
    Ho, Wo = layer.out_hw(H, W)
    y = torch.zeros((N, nt, Ho, Wo), dtype=torch.float32, device=dev) if nt else new_act(N, Ho, Wo, Cout, dtype, dev)
    yt = y if nt else y.t
    prog = make_program()
    layer.record(prog, xv, y)
    d = prog.conv_records[-1][3]
    mine, offered = hints8(hip_lib, d)
    assert mine, f"{name}: the 8-phase tile is not offered"
    d.tile_hint = 0 if not nt else next(h for h in offered if h not in mine)
    run_program(prog)
    ref = yt.float().clone()
    scale = max(1.0, ref.abs().max().item())
    for h in mine:
        d.tile_hint = h
        outs = []
        for _ in range(4):
            yt.fill_(3.0)
            run_program(prog)
            outs.append(yt.clone())
        assert all(torch.equal(outs[0], o) for o in outs[1:]), f"{name} hint {h:#x}: runs differ"
        err = (outs[0].float() - ref).abs().max().item()
        assert err <= (2e-3 if nt else 1e-2) * scale, f"{name} hint {h:#x}: differs from the default tile by {err:.3e} (scale {scale:.2f})"
