"""ft_conv_direct_fwd (1x1 convs of the deep stages as a weight-streaming GEMM, weights straight to registers) vs the CPU
oracle for the stock layers (torch CPU fp32 functional = the reference's own arithmetic, blocks.py:89-103) and vs
ft_conv2d_fwd on the same packed weights."""
import pytest
import torch
import torch.nn.functional as F

from flowtrack.pytorch_amd import hip_ops, synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv, FusedShortcutConv
from util import make_program, nchw_to_view, run_program, view_to_nchw

pytestmark = pytest.mark.gpu

# name, N, H, W, Cin, Cout, residual, ksplit the library should pick
CASES = [
    ("l4_conv3_res", 64, 8, 6, 512, 2048, True, 1),          # layer4.x.conv3 + identity residual: 32 x 8 workgroups
    ("l4_conv1", 64, 8, 6, 2048, 512, False, 4),             # layer4.x.conv1: N-tile 64, K split over the waves
    ("l3_conv1_like", 10, 16, 12, 1024, 256, False, 4),      # few pixels: K-split form
    ("ragged_pixels", 3, 7, 5, 256, 256, True, 4),           # 105 pixels: a ragged second pixel tile
    ("ragged_pixels_wide", 5, 9, 7, 320, 512, False, 1),     # K = 5 chunks (not a multiple of the ring), 315 pixels
    ("many_tiles", 64, 16, 12, 256, 1024, True, 1),          # 128 x 4 workgroups: two rounds on 256 CUs
    ("l2_entry_conv1", 24, 64, 48, 256, 128, False, 0),      # > 65536 pixels, K = 256: the persistent weight-stationary form
    ("stationary_ragged", 7, 97, 101, 256, 256, False, 0),   # 68579 pixels (ragged last tile), two channel blocks
]


def _bn(seed, name, c):
    return {"weight": synth.uniform(seed, name + "g", (c,), 0.5, 1.5), "bias": synth.normal(seed, name + "b", (c,), 0.1),
            "running_mean": synth.normal(seed, name + "m", (c,), 0.1), "running_var": synth.uniform(seed, name + "v", (c,), 0.5, 1.5),
            "eps": 1e-5}


def _bnf(y, bn):
    return F.batch_norm(y, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], training=False, eps=1e-5)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_direct_conv1x1_matches_oracle_and_igemm(hip_lib, case, monkeypatch):
    name, N, H, W, Cin, Cout, with_res, ksplit = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 31
    w = synth.normal(seed, name + ".w", (Cout, Cin, 1, 1), std=(2.0 / Cin) ** 0.5)
    bn = _bn(seed, name + ".bn", Cout)
    x = synth.normal(seed, name + ".x", (N, Cin, H, W)).half().float()
    r = synth.normal(seed, name + ".r", (N, Cout, H, W)).half().float()
    want = _bnf(F.conv2d(x, w), bn)
    want = F.relu(want + r) if with_res else F.relu(want)
    conv = FusedConv(w, bn=bn, act="relu", dtype=dtype, device=dev, label=name)
    xv = nchw_to_view(x, dtype, dev, cstride=Cin + 32, coff=32)
    rv = nchw_to_view(r, dtype, dev) if with_res else None
    outs = {}
    for mode in (True, False):
        monkeypatch.setattr(hip_ops, "CONV_DIRECT", mode)
        monkeypatch.setattr(hip_ops, "_TILE_CACHE", {})
        monkeypatch.setattr(hip_ops, "CONV_DIRECT_MAX_PIXELS", 1 << 20)
        y = ActView(torch.full((N, H, W, Cout + 64), 3.0, dtype=dtype, device=dev), Cout, 64)
        prog = make_program()
        conv.record(prog, xv, y, residual=rv)
        prog.resolve_choices()      # recorded as [direct | implicit GEMM]: keep the first form
        assert prog.calls[0][0] == ("ft_conv_direct_fwd" if mode else "ft_conv2d_fwd_ws"), prog.calls[0][0]
        run_program(prog)
        outs[mode] = view_to_nchw(y)
        assert torch.all(y.t[..., :64] == 3.0), "channels outside the output slice were written"
        if mode:
            d = prog.conv_records[0][3]
            want_bytes = Cout * Cin * 2 if ksplit == 0 else (Cout // (256 if ksplit == 1 else 64)) * (Cin // (64 if ksplit == 1 else 256)) * 32768
            assert hip_lib.ft_conv_direct_weight_bytes(d) == want_bytes
            y.t.fill_(5.0)
            run_program(prog)                      # determinism
            assert torch.equal(view_to_nchw(y), outs[True])
    scale = max(1.0, want.abs().max().item())
    err = (outs[True] - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name}: direct conv vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    diff = (outs[True] - outs[False]).abs()
    assert diff.max().item() <= 1e-2 * scale, f"{name}: direct vs igemm max abs diff {diff.max().item():.3e}"
    assert (diff > 0).float().mean().item() < 0.05, "same fp16 inputs, fp32 accumulation: only the summation order differs"


@pytest.mark.parametrize("case", [("l4_entry", 64, 16, 12, 512, 1024, 2048, 2), ("l3_entry_small", 6, 32, 24, 256, 512, 1024, 2),
                                  ("stride1", 4, 9, 7, 256, 256, 1024, 1)], ids=lambda c: c[0])
def test_direct_shortcut_conv_matches_oracle_and_igemm(hip_lib, case, monkeypatch):
    """conv3 + bn3 + projection shortcut (stride-s 1x1 conv + bn on the block input) + relu as one GEMM over K = [t2 | x]."""
    name, N, Hx, Wx, planes, cin_x, cout, s = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 33
    H, W = (Hx - 1) // s + 1, (Wx - 1) // s + 1
    w3 = synth.normal(seed, name + ".w3", (cout, planes, 1, 1), std=(2.0 / planes) ** 0.5)
    wd = synth.normal(seed, name + ".wd", (cout, cin_x, 1, 1), std=(2.0 / cin_x) ** 0.5)
    bn3, bnd = _bn(seed, name + ".bn3", cout), _bn(seed, name + ".bnd", cout)
    t2 = synth.normal(seed, name + ".t2", (N, planes, H, W)).half().float()
    x = synth.normal(seed, name + ".x", (N, cin_x, Hx, Wx)).half().float()
    want = F.relu(_bnf(F.conv2d(t2, w3), bn3) + _bnf(F.conv2d(x, wd, stride=s), bnd))
    fused = FusedShortcutConv(w3, bn3, wd, bnd, s, dtype=dtype, device=dev, act="relu", label=name)
    t2v, xv = nchw_to_view(t2, dtype, dev), nchw_to_view(x, dtype, dev)
    outs = {}
    for mode in (True, False):
        monkeypatch.setattr(hip_ops, "CONV_DIRECT", mode)
        monkeypatch.setattr(hip_ops, "_TILE_CACHE", {})
        monkeypatch.setattr(hip_ops, "CONV_DIRECT_MAX_PIXELS", 1 << 20)
        y = ActView(torch.zeros((N, H, W, cout), dtype=dtype, device=dev), cout, 0)
        prog = make_program()
        fused.record(prog, t2v, xv, y)
        prog.resolve_choices()      # recorded as [direct | implicit GEMM]: keep the first form
        assert prog.calls[0][0] == ("ft_conv_direct_fwd" if mode else "ft_conv2d_fwd"), prog.calls[0][0]
        run_program(prog)
        outs[mode] = view_to_nchw(y)
    scale = max(1.0, want.abs().max().item())
    err = (outs[True] - want).abs().max().item()
    assert err <= 3e-2 * scale, f"{name}: direct shortcut conv vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    assert (outs[True] - outs[False]).abs().max().item() <= 1e-2 * scale


@pytest.mark.parametrize("case", [("l4_conv2_8x6", 64, 8, 6), ("odd_batch", 5, 8, 6), ("r101_12x9", 7, 12, 9), ("tiny_3x5", 3, 3, 5),
                                  # maps of more than 128 pixels at 512 channels: strips of R output rows + halo rows (FlowNet's conv5_1)
                                  ("flownet_conv5_1_strips", 16, 12, 16), ("strips_ragged_last", 9, 13, 16), ("strips_w10", 9, 20, 10),
                                  ("flownet_conv6_1_c1024", 5, 6, 8, 1024), ("c1024_full_tile_8x8", 2, 8, 8, 1024), ("c1024_tiny", 3, 2, 3, 1024)],
                         ids=lambda c: c[0])
def test_direct_conv3x3_whole_maps_matches_oracle_and_igemm(hip_lib, case, monkeypatch):
    """layer4's conv2 (3x3 / stride 1 / pad 1, 512 -> 512) on whole small maps: input tile resident in LDS, taps as row
    shifts (zero row outside the image), waves split K, weights straight to registers."""
    name, N, H, W = case[:4]
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 35
    C = case[4] if len(case) > 4 else 512       # 1024: FlowNet's conv6_1 (FlowNetS.py:32): one image per workgroup, four 4-slice steps per tap
    w = synth.normal(seed, name + ".w", (C, C, 3, 3), std=(2.0 / (9 * C)) ** 0.5)
    bn = _bn(seed, name + ".bn", C)
    x = synth.normal(seed, name + ".x", (N, C, H, W)).half().float()
    want = F.relu(_bnf(F.conv2d(x, w, padding=1), bn))
    conv = FusedConv(w, pad=1, bn=bn, act="relu", dtype=dtype, device=dev, label=name)
    xv = nchw_to_view(x, dtype, dev)
    outs = {}
    for mode in (True, False):
        monkeypatch.setattr(hip_ops, "CONV_DIRECT", mode)
        monkeypatch.setattr(hip_ops, "_TILE_CACHE", {})
        y = ActView(torch.full((N, H, W, C + 32), 3.0, dtype=dtype, device=dev), C, 32)
        prog = make_program()
        conv.record(prog, xv, y)
        prog.resolve_choices()      # recorded as [direct | implicit GEMM]: keep the first form
        assert prog.calls[0][0] == ("ft_conv_direct_fwd" if mode else "ft_conv2d_fwd_ws"), prog.calls[0][0]
        run_program(prog)
        outs[mode] = view_to_nchw(y)
        assert torch.all(y.t[..., :32] == 3.0)
    scale = max(1.0, want.abs().max().item())
    err = (outs[True] - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name}: direct 3x3 vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    diff = (outs[True] - outs[False]).abs()
    assert diff.max().item() <= 1e-2 * scale and (diff > 0).float().mean().item() < 0.05


# name, N, H, W, Cin, Cout, stride, act, input channel offset
GATHER_CASES = [
    ("l3_entry_conv2", 64, 32, 24, 256, 256, 2, "relu", 0),      # layer3.0.conv2: K-split form, 9 chunks, 512 workgroups
    ("l4_entry_conv2", 64, 16, 12, 512, 512, 2, "relu", 0),      # layer4.0.conv2: 18 chunks
    ("flownet_conv4", 4, 48, 64, 256, 512, 2, "leaky", 32),      # N-tile 256 form, 36 chunks of 64 channels, LeakyReLU, input slice
    ("stride1_big_map", 2, 20, 14, 256, 256, 1, "relu", 0),      # 3x3 / s1 on a map the whole-map kernel does not take
    ("odd_map_s2", 3, 13, 9, 256, 256, 2, "relu", 0),            # odd sizes: Ho = 7, Wo = 5, ragged pixel tile
    ("long_walk", 2, 8, 6, 1024, 256, 2, "relu", 0),             # 36 chunks of 256 channels: the run-time loop
    # 512 -> * / stride 2 on input maps of <= 256 pixels: the whole-map stride-2 form (conv3x3s2_direct_kernel: two passes of 256
    # channels over a resident input map) — l4_entry_conv2 above takes it as well
    ("s2_whole_odd_map", 3, 13, 9, 512, 512, 2, "relu", 0),      # Ho x Wo = 7 x 5: ragged second pixel tile, odd input rows / columns
    ("s2_whole_flownet_conv6", 4, 12, 16, 512, 1024, 2, "leaky", 32),   # FlowNetS conv6: 16 channel blocks, LeakyReLU, input slice
    ("s2_whole_one_tile", 5, 8, 6, 512, 128, 2, "relu", 0),      # Ho x Wo = 4 x 3: one pixel tile (MT = 1)
    ("s2_whole_16x16", 2, 16, 16, 512, 64, 2, "relu", 0),        # the largest input map (256 pixels), 64 output pixels exactly
    # enough (image pair, channel block) workgroups for the TWO-images-per-workgroup form (conv3x3s2p_direct_kernel: four passes of
    # 128 channels, K split by channel half x tap parity) — l4_entry_conv2 above takes it as well (32 pairs x 8 blocks)
    ("s2_pair_odd_batch", 51, 12, 16, 512, 512, 2, "leaky", 32), # the last workgroup's second image does not exist; input slice
    ("s2_pair_small_map", 50, 8, 8, 512, 512, 2, "relu", 0),     # 2 x 16 output pixels: one of the kernel's two pixel tiles is empty
    ("s2_pair_odd_map", 56, 13, 9, 512, 512, 2, "relu", 0),      # Ho x Wo = 7 x 5 per image: 70 output pixels, ragged third tile
    # input maps of more than 256 pixels: strips of R output rows over their 2 R + 1 input rows (FlowNetS conv5: 24 x 32 -> 12 x 16, R = 3)
    ("s2_strips_flownet_conv5", 16, 24, 32, 512, 512, 2, "leaky", 0),
    ("s2_strips_ragged", 9, 26, 20, 512, 512, 2, "relu", 32),    # Ho = 13 = 5 + 5 + 3 rows, odd input height, input slice
]


@pytest.mark.parametrize("case", GATHER_CASES, ids=[c[0] for c in GATHER_CASES])
def test_direct_conv3x3_gather_matches_oracle_and_igemm(hip_lib, case, monkeypatch):
    """3x3 convs (stride 1 or 2, pad 1; the entry blocks' conv2 with the stride on the 3x3, blocks.py:92-95, and FlowNetS's
    conv4 .. conv6, FlowNetS.py:24-31) on the weight-streaming kernel: the pixel operand of a tap is a gather."""
    name, N, H, W, Cin, Cout, stride, act, xoff = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 37
    w = synth.normal(seed, name + ".w", (Cout, Cin, 3, 3), std=(2.0 / (9 * Cin)) ** 0.5)
    bn = _bn(seed, name + ".bn", Cout)
    x = synth.normal(seed, name + ".x", (N, Cin, H, W)).half().float()
    pre = _bnf(F.conv2d(x, w, stride=stride, padding=1), bn)
    want = F.relu(pre) if act == "relu" else F.leaky_relu(pre, 0.1)
    Ho, Wo = want.shape[2], want.shape[3]
    conv = FusedConv(w, stride=stride, pad=1, bn=bn, act=act, slope=0.1, dtype=dtype, device=dev, label=name)
    xv = nchw_to_view(x, dtype, dev, cstride=Cin + xoff, coff=xoff)
    if xoff:
        xv.t[..., :xoff] = 7.0
    outs = {}
    for mode in (True, False):
        monkeypatch.setattr(hip_ops, "CONV_DIRECT", mode)
        monkeypatch.setattr(hip_ops, "_TILE_CACHE", {})
        y = ActView(torch.full((N, Ho, Wo, Cout + 32), 3.0, dtype=dtype, device=dev), Cout, 32)
        prog = make_program()
        conv.record(prog, xv, y)
        prog.resolve_choices()      # recorded as [direct | implicit GEMM]: keep the first form
        assert prog.calls[0][0] == ("ft_conv_direct_fwd" if mode else "ft_conv2d_fwd_ws"), prog.calls[0][0]
        run_program(prog)
        outs[mode] = view_to_nchw(y)
        assert torch.all(y.t[..., :32] == 3.0)
        if mode:
            y.t.fill_(5.0)
            run_program(prog)
            assert torch.equal(view_to_nchw(y), outs[True])
    scale = max(1.0, want.abs().max().item())
    err = (outs[True] - want).abs().max().item()
    assert err <= 2e-2 * scale, f"{name}: direct 3x3 gather vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    diff = (outs[True] - outs[False]).abs()
    assert diff.max().item() <= 1e-2 * scale and (diff > 0).float().mean().item() < 0.05


def test_stationary_k512_dev_form(hip_lib):
    """The 8-wave K = 512 weight-stationary form is a dev alternative (FT_CD_STATIONARY=2, read once per process): checked in
    a subprocess against the torch-CPU functional form."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch, torch.nn.functional as F
sys.path.insert(0, "tests")
from flowtrack.pytorch_amd import hip_ops, synth
from flowtrack.pytorch_amd.hip_ops import ActView, FusedConv
from util import make_program, nchw_to_view, run_program, view_to_nchw
for name, N, H, W in (("a", 64, 32, 24), ("b", 3, 111, 101)):
    w = synth.normal(41, name + ".w", (256, 512, 1, 1), std=(2.0 / 512) ** 0.5)
    x = synth.normal(41, name + ".x", (N, 512, H, W)).half().float()
    want = F.relu(F.conv2d(x, w))
    conv = FusedConv(w, act="relu", dtype=torch.float16, device=torch.device("cuda:0"), label=name)
    xv = nchw_to_view(x, torch.float16, torch.device("cuda:0"))
    y = ActView(torch.zeros((N, H, W, 256), dtype=torch.float16, device="cuda:0"), 256, 0)
    prog = make_program(); conv.record(prog, xv, y); prog.resolve_choices()
    assert prog.calls[0][0] == "ft_conv_direct_fwd", prog.calls[0][0]
    d = prog.conv_records[0][3]
    assert hip_ops._lib.load().ft_conv_direct_weight_bytes(d) == 256 * 512 * 2      # the stationary form's stream, not the K-split one's
    run_program(prog)
    err = (view_to_nchw(y) - want).abs().max().item()
    assert err <= 2e-2 * max(1.0, want.abs().max().item()), err
print("ok")
'''
    env = dict(os.environ, FT_CD_STATIONARY="2", FT_CONV_DIRECT_MAX_PIXELS="1048576")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("case", [("flownet_conv2_like", 2, 96, 128, 128, "bias"),     # whole 8 x 8 patches, two output groups
                                  ("ragged_patches", 3, 41, 55, 128, "bn"),            # Ho x Wo = 21 x 28: ragged last patch row / column
                                  ("one_group", 2, 30, 34, 64, "bias"),                # Cout = 64: one group, odd sizes
                                  ("three_groups", 1, 64, 48, 192, "none"),            # 3 does not divide an XCD's 32 workgroups
                                  ("tiny_map", 5, 5, 3, 256, "bn"),                    # a map smaller than one patch, four groups
                                  ("many_patches", 7, 136, 200, 128, "bias")],         # 7 x 9 x 13 = 819 patches: > 6 per workgroup pair
                         ids=lambda c: c[0])
def test_direct_conv5x5s2_register_stationary_matches_oracle_and_igemm(hip_lib, case, monkeypatch):
    """FlowNet's conv2 (5x5 / stride 2 / pad 2 on 64 channels, FlowNetS.py:21): weights stationary in registers, 19 x 19 input
    patches double-buffered in LDS, K quarters reduced through LDS (conv_wstat.hip)."""
    name, N, H, W, Cout, norm = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 37
    Cin = 64
    w = synth.normal(seed, name + ".w", (Cout, Cin, 5, 5), std=(2.0 / (25 * Cin)) ** 0.5)
    bias = synth.normal(seed, name + ".bias", (Cout,), 0.2) if norm == "bias" else None
    bn = _bn(seed, name + ".bn", Cout) if norm == "bn" else None
    x = synth.normal(seed, name + ".x", (N, Cin, H, W)).half().float()
    want = F.conv2d(x, w, bias, stride=2, padding=2)
    if bn is not None:
        want = _bnf(want, bn)
    want = F.leaky_relu(want, 0.1)
    conv = FusedConv(w, stride=2, pad=2, bias=bias, bn=bn, act="leaky", slope=0.1, dtype=dtype, device=dev, label=name)
    xv = nchw_to_view(x, dtype, dev, cstride=Cin + 16, coff=8)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    outs = {}
    for mode in (True, False):
        monkeypatch.setattr(hip_ops, "CONV_DIRECT", mode)
        monkeypatch.setattr(hip_ops, "_TILE_CACHE", {})
        y = ActView(torch.full((N, Ho, Wo, Cout + 24), 3.0, dtype=dtype, device=dev), Cout, 16)
        prog = make_program()
        conv.record(prog, xv, y)
        prog.resolve_choices()      # recorded as [direct | implicit GEMM]: keep the first form
        assert prog.calls[0][0] == ("ft_conv_direct_fwd" if mode else "ft_conv2d_fwd_ws"), prog.calls[0][0]
        run_program(prog)
        outs[mode] = view_to_nchw(y)
        assert torch.all(y.t[..., :16] == 3.0) and torch.all(y.t[..., 16 + Cout:] == 3.0), "channels outside the output slice were written"
        if mode:
            d = prog.conv_records[0][3]
            assert hip_lib.ft_conv_direct_weight_bytes(d) == (Cout // 64) * 200 * 1024
            for _ in range(3):                     # determinism (the exchange buffer and both patch buffers are recycled)
                y.t.fill_(5.0)
                run_program(prog)
                assert torch.equal(view_to_nchw(y), outs[True])
    scale = max(1.0, want.abs().max().item())
    err = (outs[True] - want).abs().max().item()
    assert err <= 1e-2 * scale, f"{name}: register-stationary conv vs oracle max abs err {err:.3e} (scale {scale:.2f})"
    diff = (outs[True] - outs[False]).abs()
    assert diff.max().item() <= 5e-3 * scale, f"{name}: direct vs igemm max abs diff {diff.max().item():.3e}"
    assert (diff > 0).float().mean().item() < 0.05, "same fp16 inputs, fp32 accumulation: only the summation order differs"
