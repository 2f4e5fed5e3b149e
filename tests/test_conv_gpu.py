"""HIP fused conv / transposed conv vs the CPU oracle for the stock layers (torch CPU fp32 functional —
the arithmetic the reference itself runs for nn.Conv2d / ConvTranspose2d / BatchNorm2d),
launched through the C ABI (ft_conv2d_fwd).  Shapes are the distinct layer kinds of SURVEY Appendix A."""
import pytest
import torch
import torch.nn.functional as F

from flowtrack.pytorch_amd import synth
from flowtrack.pytorch_amd.hip_ops import ActView, FlowtrackHipError, FusedConv, act_stride
from util import make_program, nchw_to_view, run_program, view_to_nchw

pytestmark = pytest.mark.gpu

# (name, N, Cin, H, W, Cout, k, stride, pad, transposed, bias, bn, act, residual)
CASES = [
    ("1x1_64_256", 2, 64, 16, 12, 256, 1, 1, 0, False, False, True, "relu", False),
    ("1x1_res_relu", 2, 128, 16, 12, 512, 1, 1, 0, False, False, True, "relu", True),
    ("1x1_s2_downsample", 2, 256, 16, 12, 512, 1, 2, 0, False, False, True, None, False),
    ("3x3_s1", 2, 64, 16, 12, 64, 3, 1, 1, False, False, True, "relu", False),
    ("3x3_s2", 3, 128, 16, 12, 128, 3, 2, 1, False, False, True, "relu", False),
    ("3x3_ragged_m", 1, 64, 7, 5, 64, 3, 1, 1, False, False, True, "relu", False),
    ("stem_7x7_cin3", 2, 3, 64, 48, 64, 7, 2, 3, False, False, True, "relu", False),
    ("flow_7x7_cin6_leaky", 1, 6, 64, 64, 64, 7, 2, 3, False, True, False, "leaky", False),
    ("flow_5x5_s2", 1, 64, 32, 32, 128, 5, 2, 2, False, True, False, "leaky", False),
    ("flow_cin12", 1, 12, 32, 32, 64, 7, 2, 3, False, True, False, "leaky", False),
    ("predict_flow_cin194", 1, 194, 16, 24, 2, 3, 1, 1, False, True, False, None, False),
    ("conv_cin473", 1, 473, 8, 12, 256, 3, 1, 1, False, True, False, "leaky", False),
    ("conv_redir_cout32", 1, 256, 8, 12, 32, 1, 1, 0, False, True, False, "leaky", False),
    ("heatmap_cout17", 2, 256, 16, 12, 17, 1, 1, 0, False, True, False, None, False),
    ("deconv_256", 2, 256, 8, 6, 256, 4, 2, 1, True, False, True, "relu", False),
    ("deconv_2048_k8192", 1, 2048, 4, 3, 256, 4, 2, 1, True, False, True, "relu", False),
    ("deconv_cin1026_bias_leaky", 1, 1026, 4, 6, 256, 4, 2, 1, True, True, False, "leaky", False),
    ("upflow_2_2", 1, 2, 6, 8, 2, 4, 2, 1, True, False, False, None, False),
    ("upflow_2_2_bias", 1, 2, 6, 8, 2, 4, 2, 1, True, True, False, None, False),
    ("3x3_cout1024_k4608", 1, 512, 6, 8, 1024, 3, 2, 1, False, True, False, "leaky", False),
    ("3x3_m_gt_tiles", 5, 64, 24, 20, 192, 3, 1, 1, False, False, True, "relu", True),
    ("5x5_s2_cin128", 2, 128, 24, 32, 256, 5, 2, 2, False, True, False, "leaky", False),
    ("1x1_cin2048", 3, 2048, 8, 6, 512, 1, 1, 0, False, False, True, "relu", False),
    ("deconv_770_cout128", 1, 770, 6, 8, 128, 4, 2, 1, True, True, False, "leaky", False),
    ("predict_flow_cin1026", 2, 1026, 6, 8, 2, 3, 1, 1, False, True, False, None, False),
    # >= 256 patches of 8x16: the LDS-patch predict_flow kernel (conv_pflow_kernel), 3.03 chunks of 64 channels / ragged, Cout 1
    ("predict_flow_patch_cin194", 4, 194, 64, 128, 2, 3, 1, 1, False, True, False, None, False),
    ("predict_flow_patch_ragged_cout1", 3, 40, 70, 150, 1, 3, 1, 1, False, True, False, "leaky", False),
    # >= 256 tiles of 12x16: the matrix-pipe predict_flow kernel (conv_pflow_mfma_kernel): 7 chunks of 32 channels with a ragged
    # last one, ragged image edges + Cout 1 + activation, and predict_flow3's 386 channels (13 chunks: ring wraps three times)
    ("predict_flow_mfma_cin194", 6, 194, 64, 128, 2, 3, 1, 1, False, True, False, None, False),
    ("predict_flow_mfma_ragged_cout1", 5, 40, 70, 150, 1, 3, 1, 1, False, True, False, "leaky", False),
    ("predict_flow_mfma_cin386", 16, 386, 48, 64, 2, 3, 1, 1, False, True, False, None, False),
    ("fewout_cout3_5x5_s2", 1, 40, 11, 9, 3, 5, 2, 2, False, True, False, "leaky", False),
    ("fewout_cout4_1x1_res", 2, 64, 7, 5, 4, 1, 1, 0, False, False, True, "relu", True),
]


def _reference(x, w, bias, bn, stride, pad, transposed, act, res):
    if transposed:
        y = F.conv_transpose2d(x, w, bias, stride=stride, padding=pad)
    else:
        y = F.conv2d(x, w, bias, stride=stride, padding=pad)
    if bn is not None:
        y = F.batch_norm(y, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], training=False, eps=1e-5)
    if res is not None:
        y = y + res
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, 0.1)
    return y


@pytest.mark.parametrize("layout", ["wide", "tight"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_matches_oracle(hip_lib, case, dtype, layout):
    """layout 'wide': channel stride as the networks allocate it (act_stride: multiples of 32 -> the
    direct-to-LDS kernel whenever Cout > 32); 'tight': stride roundup8(Cin) -> the generic kernel for
    ragged channel counts.  Both must agree with the oracle."""
    name, N, Cin, H, W, Cout, k, stride, pad, transposed, has_bias, has_bn, act, has_res = case
    dev = torch.device("cuda:0")
    seed = 11
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    fan = (Cin * 4) if transposed else Cin * k * k
    w = synth.normal(seed, name + ".w", wshape, std=(2.0 / fan) ** 0.5)
    x = synth.normal(seed, name + ".x", (N, Cin, H, W))
    bias = synth.normal(seed, name + ".b", (Cout,), std=0.2) if has_bias else None
    bn = None
    if has_bn:
        bn = {"weight": synth.uniform(seed, name + ".g", (Cout,), 0.5, 1.5), "bias": synth.normal(seed, name + ".be", (Cout,), 0.1),
              "running_mean": synth.normal(seed, name + ".m", (Cout,), 0.1), "running_var": synth.uniform(seed, name + ".v", (Cout,), 0.5, 1.5),
              "eps": 1e-5}
    if dtype == torch.float16:  # the oracle sees the same fp16-rounded operands, accumulates in fp32
        w, x = w.half().float(), x.half().float()
    layer = FusedConv(w, dtype=dtype, device=dev, stride=stride, pad=pad, transposed=transposed, bias=bias, bn=bn, act=act,
                      slope=0.1, label=name)
    Ho, Wo = layer.out_hw(H, W)
    res = None
    if has_res:
        res = synth.normal(seed, name + ".r", (N, Cout, Ho, Wo))
        if dtype == torch.float16:
            res = res.half().float()
    want = _reference(x, w, bias, bn, stride, pad, transposed, act, res)

    xv = nchw_to_view(x, dtype, dev, cstride=act_stride(Cin) if layout == "wide" else None)
    # write into a channel slice of a wider buffer whose other channels must stay untouched
    ycs = ((Cout + 8 + 7) // 8) * 8 + 8
    ybuf = torch.full((N, Ho, Wo, ycs), 7.0, dtype=dtype, device=dev)
    yv = ActView(ybuf, Cout, 8)
    prog = make_program()
    layer.record(prog, xv, yv, residual=nchw_to_view(res, dtype, dev) if res is not None else None)
    # and the NCHW fp32 output form
    y_nchw = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=dev)
    layer.record(prog, xv, y_nchw, residual=nchw_to_view(res, dtype, dev) if res is not None else None)
    run_program(prog)

    got = view_to_nchw(yv)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    err_nchw = (y_nchw.cpu() - want).abs().max().item()
    assert err <= tol * scale, f"{name} {dtype}: NHWC max abs err {err:.3e} (scale {scale:.2f})"
    assert err_nchw <= (2e-4 if dtype == torch.float32 else 2e-3) * scale, f"{name} {dtype}: NCHW max abs err {err_nchw:.3e}"
    # guard channels around the slice untouched
    assert torch.all(ybuf[..., :8] == 7.0) and torch.all(ybuf[..., 8 + Cout:] == 7.0), "wrote outside its channel slice"


NONFINITE = [
    # name, N, Cin, H, W, Cout, k, stride, pad: one shape per epilogue family that implements ReLU as max(v, 0 (*) v)
    ("igemm_3x3_64", 2, 64, 16, 12, 64, 3, 1, 1),               # conv_igemm_dma_kernel (conv_common.h: apply_act)
    ("direct_1x1_cin2048", 64, 2048, 8, 6, 512, 1, 1, 0),       # layer4.x.conv1 at the benchmarked batch: conv_direct_kernel
    ("direct_1x1_cout2048", 64, 512, 8, 6, 2048, 1, 1, 0),      # layer4.x.conv3 (without its residual)
    ("direct_3x3_512", 64, 512, 8, 6, 512, 3, 1, 1),            # layer4.x.conv2: conv3x3_direct_kernel
]


@pytest.mark.parametrize("case", NONFINITE, ids=[c[0] for c in NONFINITE])
def test_relu_of_non_finite_values(hip_lib, case):
    """ReLU of a pre-activation that is -inf / +inf / NaN: the epilogues computed max(v, k * v) with k = 0, and
    the IEEE product -inf * 0 = NaN turned ReLU(-inf) into -inf (torch: 0), which then rode every following residual.  Now
    `!(v <= 0) ? v : k (*) v` with the legacy multiply (0 * x = 0): ft_common.h, act_mul.  The non-finite values enter through the bias (a conv sum with an infinite input
    would be NaN in the other channels by itself): channel 1 gets -inf, channel 2 +inf, channel 3 NaN (stays NaN, as in torch), every
    other channel must still match the oracle."""
    name, N, Cin, H, W, Cout, k, stride, pad = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 37
    w = synth.normal(seed, name + ".w", (Cout, Cin, k, k), std=(2.0 / (Cin * k * k)) ** 0.5).half().float()
    x = synth.normal(seed, name + ".x", (N, Cin, H, W)).half().float()
    bias = synth.normal(seed, name + ".b", (Cout,), std=0.2)
    bias[1], bias[2], bias[3] = float("-inf"), float("inf"), float("nan")
    layer = FusedConv(w, dtype=dtype, device=dev, stride=stride, pad=pad, bias=bias, act="relu", label=name)
    want = F.relu(F.conv2d(x, w, bias, stride=stride, padding=pad))
    xv = nchw_to_view(x, dtype, dev, cstride=act_stride(Cin))
    Ho, Wo = layer.out_hw(H, W)
    yv = ActView(torch.full((N, Ho, Wo, act_stride(Cout)), 7.0, dtype=dtype, device=dev), Cout, 0)
    prog = make_program()
    layer.record(prog, xv, yv)
    run_program(prog)
    got = view_to_nchw(yv)
    print(f"{name}: launched {[c[0] for c in prog.calls]}")
    assert torch.all(got[:, 1] == 0.0), f"{name}: ReLU(-inf) must be 0, got {got[:, 1].flatten()[:4].tolist()}"
    assert torch.all(torch.isposinf(got[:, 2])), f"{name}: ReLU(+inf) must stay +inf"
    assert torch.all(torch.isnan(got[:, 3])), f"{name}: ReLU(NaN) must stay NaN (F.relu does the same)"
    keep = [c for c in range(Cout) if c not in (1, 2, 3)]
    scale = max(1.0, want[:, keep].abs().max().item())
    assert (got[:, keep] - want[:, keep]).abs().max().item() <= 2e-2 * scale


# ---- row-packed small-Cin stems (7x7/s2 on 3/6/12 channels): whole kernel rows as K-runs -----------------
ROWPACK = [
    # name, N, Cin, H, W, Cout, k, stride, pad, bias, bn, act
    ("pose_stem_3", 2, 3, 64, 48, 64, 7, 2, 3, False, True, "relu"),
    ("flow_conv1_6", 1, 6, 64, 64, 64, 7, 2, 3, True, False, "leaky"),
    ("flow_conv1_12", 1, 12, 32, 64, 64, 7, 2, 3, True, False, "leaky"),
    # >= 512 tiles of 8 x 16: the persistent weight-stationary stem (conv_stem_persist_kernel), every row width (64 / 128 / 256
    # bytes per kernel row), ragged right / bottom edges, 2-3 tiles per workgroup and a batch that wraps images inside a range
    ("persist_pose_stem_3", 8, 3, 256, 192, 64, 7, 2, 3, False, True, "relu"),
    ("persist_flow_conv1_6_ragged", 3, 6, 250, 500, 64, 7, 2, 3, True, False, "leaky"),
    ("persist_flow_conv1_12_ragged", 4, 12, 250, 260, 64, 7, 2, 3, True, False, "leaky"),
    ("ragged_5x5_s1", 2, 3, 9, 11, 128, 5, 1, 2, True, False, None),
    ("stem_3x3_s2_cin4", 1, 4, 10, 14, 40, 3, 2, 1, False, True, "relu"),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("case", ROWPACK, ids=[c[0] for c in ROWPACK])
def test_rowpacked_conv_matches_oracle(hip_lib, case, dtype):
    from flowtrack.pytorch_amd.hip_ops import new_rowpacked_act
    name, N, Cin, H, W, Cout, k, stride, pad, has_bias, has_bn, act = case
    dev = torch.device("cuda:0")
    seed = 17
    w = synth.normal(seed, name + ".w", (Cout, Cin, k, k), std=(2.0 / (Cin * k * k)) ** 0.5)
    x = synth.normal(seed, name + ".x", (N, Cin, H, W))
    bias = synth.normal(seed, name + ".b", (Cout,), std=0.2) if has_bias else None
    bn = None
    if has_bn:
        bn = {"weight": synth.uniform(seed, name + ".g", (Cout,), 0.5, 1.5), "bias": synth.normal(seed, name + ".be", (Cout,), 0.1),
              "running_mean": synth.normal(seed, name + ".m", (Cout,), 0.1), "running_var": synth.uniform(seed, name + ".v", (Cout,), 0.5, 1.5),
              "eps": 1e-5}
    if dtype == torch.float16:
        w, x = w.half().float(), x.half().float()
    want = _reference(x, w, bias, bn, stride, pad, False, act, None)
    layer = FusedConv(w, dtype=dtype, device=dev, stride=stride, pad=pad, bias=bias, bn=bn, act=act, slope=0.1, label=name)
    xv = new_rowpacked_act(N, H, W, Cin, pad, dtype, dev)
    prog = make_program()
    # fill through the library's own packer (ft_pack_nchw_to_nhwc with lpad / wpitch), as the networks do
    from flowtrack.pytorch_amd.hip_ops import record_pack_input
    gx = x.to(dev)
    record_pack_input(prog, gx, xv)
    Ho, Wo = layer.out_hw(H, W)
    yv = ActView(torch.zeros((N, Ho, Wo, act_stride(Cout)), dtype=dtype, device=dev), Cout, 0)
    layer.record(prog, xv, yv)
    run_program(prog)
    assert torch.all(xv.t[:, :, :xv.lpad] == 0) and torch.all(xv.t[:, :, xv.lpad + W:] == 0) and torch.all(xv.t[..., Cin:] == 0)
    got = view_to_nchw(yv)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err <= tol * scale, f"{name} {dtype}: max abs err {err:.3e} (scale {scale:.2f})"


# ---- every tile variant the library offers (ft_conv_tile_candidates) must give the oracle's result ----------
VARIANT_CASES = [
    # (name, N, Cin, H, W, Cout, k, stride, pad, transposed, residual)
    ("var_3x3_256", 3, 256, 16, 12, 256, 3, 1, 1, False, False),
    ("var_1x1_res", 2, 128, 17, 13, 512, 1, 1, 0, False, True),
    ("var_3x3_s2_ragged", 2, 128, 15, 11, 128, 3, 2, 1, False, False),
    ("var_deconv_512", 2, 512, 8, 6, 256, 4, 2, 1, True, False),
    ("var_1x1_cin64", 2, 64, 16, 12, 256, 1, 1, 0, False, True),
    ("var_stem_rowpack", 2, 3, 64, 48, 64, 7, 2, 3, False, False),
    # halo-patch kernel: ragged patches, one full + one half 64-channel chunk, half chunk only, BC = 64, many chunks
    ("var_3x3_ragged_cin96", 2, 96, 17, 13, 128, 3, 1, 1, False, False),
    ("var_3x3_cin32_cout64", 1, 32, 20, 37, 64, 3, 1, 1, False, False),
    ("var_3x3_tall", 1, 64, 40, 9, 192, 3, 1, 1, False, False),
    ("var_deconv_cin1026", 1, 1026, 5, 7, 256, 4, 2, 1, True, False),
    ("var_deconv_cout64_wide_img", 1, 128, 6, 33, 64, 4, 2, 1, True, False),
    # few pixels + long K: the cross-workgroup split-K variants (workspace + reduce launch), with residual / transposed
    ("var_splitk_3x3_512", 2, 512, 8, 6, 512, 3, 1, 1, False, False),
    ("var_splitk_1x1_res_k2048", 2, 2048, 8, 6, 512, 1, 1, 0, False, True),
    ("var_splitk_deconv_1024", 1, 1024, 6, 8, 512, 4, 2, 1, True, False),
    # stem patch kernel: every row-packed stem of the networks (pose 3 ch, FlowNetS/C 6 / 3 ch, stacked nets 12 ch,
    # FlowNetSD 3x3 on 6 ch, FlowNetFusion 3x3 on 11 ch), ragged tiles
    ("var_stem_flow6", 1, 6, 64, 96, 64, 7, 2, 3, False, False),
    ("var_stem_cs12_ragged", 2, 12, 50, 70, 64, 7, 2, 3, False, False),
    ("var_stem_pose_ragged", 3, 3, 36, 28, 64, 7, 2, 3, False, False),
    ("var_stem_sd_3x3", 1, 6, 24, 40, 64, 3, 1, 1, False, False),
    ("var_stem_fusion_3x3", 2, 11, 19, 33, 64, 3, 1, 1, False, False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("case", VARIANT_CASES, ids=[c[0] for c in VARIANT_CASES])
def test_every_tile_variant_matches_oracle(hip_lib, case, dtype):
    import ctypes
    from flowtrack.pytorch_amd.hip_ops import new_rowpacked_act
    name, N, Cin, H, W, Cout, k, stride, pad, transposed, has_res = case
    dev = torch.device("cuda:0")
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    fan = (Cin * 4) if transposed else Cin * k * k
    w = synth.normal(5, name + ".w", wshape, std=(2.0 / fan) ** 0.5)
    x = synth.normal(5, name + ".x", (N, Cin, H, W))
    bn = {"weight": synth.uniform(5, name + ".g", (Cout,), 0.5, 1.5), "bias": synth.normal(5, name + ".be", (Cout,), 0.1),
          "running_mean": synth.normal(5, name + ".m", (Cout,), 0.1), "running_var": synth.uniform(5, name + ".v", (Cout,), 0.5, 1.5),
          "eps": 1e-5}
    if dtype == torch.float16:
        w, x = w.half().float(), x.half().float()
    layer = FusedConv(w, dtype=dtype, device=dev, stride=stride, pad=pad, transposed=transposed, bn=bn, act="relu", label=name)
    Ho, Wo = layer.out_hw(H, W)
    res = synth.normal(5, name + ".r", (N, Cout, Ho, Wo)) if has_res else None
    if res is not None and dtype == torch.float16:
        res = res.half().float()
    want = _reference(x, w, None, bn, stride, pad, transposed, "relu", res)
    if Cin <= 16:
        xv = new_rowpacked_act(N, H, W, Cin, pad, dtype, dev)
        xv.t[:, :, xv.lpad:xv.lpad + W, :Cin] = x.permute(0, 2, 3, 1).to(device=dev, dtype=dtype)
    else:
        xv = nchw_to_view(x, dtype, dev, cstride=act_stride(Cin))
    yv = ActView(torch.zeros((N, Ho, Wo, act_stride(Cout)), dtype=dtype, device=dev), Cout, 0)
    prog = make_program()
    layer.record(prog, xv, yv, residual=nchw_to_view(res, dtype, dev) if res is not None else None)
    d = prog.conv_records[0][3]
    hints = (ctypes.c_int * 64)()
    n = hip_lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 64)
    assert n >= 2, f"{name}: only {n} tile variants offered"
    if "splitk" in name:
        assert any((int(v) >> 21) & 7 for v in hints[:n]), f"{name}: no split-K variant offered"
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    scale = max(1.0, want.abs().max().item())
    for h in [0] + [int(v) for v in hints[:n]]:
        d.tile_hint = h
        yv.t.fill_(3.0)
        torch.cuda.synchronize()      # the fill runs on torch's stream, the program on its own
        run_program(prog)
        err = (view_to_nchw(yv) - want).abs().max().item()
        assert err <= tol * scale, (f"{name} {dtype} tile bp {h & 0xfff} bc {(h >> 12) & 0x1ff} splitK {(1, 2, 4, 8, 3, 5, 6, 7)[(h >> 21) & 7]} ks {(h >> 24) & 0xf} "
                                    f"wide {(h >> 28) & 3} halo {(h >> 30) & 1}: max abs err {err:.3e}")
    assert torch.all(yv.t[..., Cout:] == 3.0) or act_stride(Cout) == Cout


# ---- conv3 + projection shortcut as one K-concatenated GEMM (ft_conv_desc.x2_*, FusedShortcutConv) -----------
SHORTCUT_CASES = [
    # (name, N, planes (conv3 input channels), Cin of the block input, H, W of the block input, stride of the shortcut)
    ("layer1_like", 2, 64, 64, 16, 12, 1),
    ("layer2_like_s2", 2, 128, 256, 16, 12, 2),
    ("layer3_like_s2_ragged", 3, 256, 512, 9, 7, 2),
    ("odd_channels", 1, 96, 160, 10, 6, 2),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("case", SHORTCUT_CASES, ids=[c[0] for c in SHORTCUT_CASES])
def test_fused_shortcut_conv_matches_oracle(hip_lib, case, dtype):
    """relu(bn3(conv3(t2)) + bn_d(conv_d(x))) (blocks.py:104-119) in one launch, every tile variant the library offers."""
    import ctypes
    from flowtrack.pytorch_amd.hip_ops import FusedShortcutConv
    name, N, planes, cin, H, W, s = case
    dev = torch.device("cuda:0")
    cout = planes * 4
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    w3 = synth.normal(9, name + ".w3", (cout, planes, 1, 1), std=(2.0 / planes) ** 0.5)
    wd = synth.normal(9, name + ".wd", (cout, cin, 1, 1), std=(2.0 / cin) ** 0.5)
    t2 = synth.normal(9, name + ".t2", (N, planes, Ho, Wo))
    x = synth.normal(9, name + ".x", (N, cin, H, W))
    mkbn = lambda tag: {"weight": synth.uniform(9, name + tag + "g", (cout,), 0.5, 1.5), "bias": synth.normal(9, name + tag + "b", (cout,), 0.1),
                        "running_mean": synth.normal(9, name + tag + "m", (cout,), 0.1),
                        "running_var": synth.uniform(9, name + tag + "v", (cout,), 0.5, 1.5), "eps": 1e-5}
    bn3, bnd = mkbn(".bn3"), mkbn(".bnd")
    if dtype == torch.float16:
        t2, x = t2.half().float(), x.half().float()
    want = F.relu(_reference(t2, w3, None, bn3, 1, 0, False, None, None) + _reference(x, wd, None, bnd, s, 0, False, None, None))
    layer = FusedShortcutConv(w3, bn3, wd, bnd, s, dtype=dtype, device=dev, label=name)
    t2v = nchw_to_view(t2, dtype, dev, cstride=act_stride(planes))
    xv = nchw_to_view(x, dtype, dev, cstride=act_stride(cin))
    yv = ActView(torch.zeros((N, Ho, Wo, act_stride(cout)), dtype=dtype, device=dev), cout, 0)
    prog = make_program()
    layer.record(prog, t2v, xv, yv)
    d = prog.conv_records[0][3]
    hints = (ctypes.c_int * 64)()
    n = hip_lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 64)
    assert n >= 2
    # fp16: BatchNorm is folded into fp16 weights here (one more rounding than scale-after-accumulate)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    scale = max(1.0, want.abs().max().item())
    for h in [0] + [int(v) for v in hints[:n]]:
        assert h == 0 or ((h >> 24) & 0xf == 1 and not (h >> 30) & 1), "K-concat offers no split-K / halo variants"
        d.tile_hint = h
        yv.t.fill_(3.0)
        run_program(prog)
        err = (view_to_nchw(yv) - want).abs().max().item()
        assert err <= tol * scale, f"{name} {dtype} hint {h:#x}: max abs err {err:.3e}"


# ---- conv / deconv + fused tail 1x1 conv (ft_conv_desc.tail_cout): the pose head's deconv -> heatmap pair ------
TAIL_CASES = [
    # (name, N, Cin, H, W, Cout, k, stride, pad, transposed, tail_cout)
    ("deconv256_heatmap17", 2, 256, 8, 6, 256, 4, 2, 1, True, 17),
    ("deconv128_tail32_ragged", 1, 64, 5, 7, 128, 4, 2, 1, True, 32),
    ("conv3x3_64_tail5", 2, 64, 9, 11, 64, 3, 1, 1, False, 5),
    ("conv1x1_256_tail1", 1, 512, 12, 10, 256, 1, 1, 0, False, 1),
]


@pytest.mark.parametrize("nchw", [True, False], ids=["nchw_f32", "nhwc_f16"])
@pytest.mark.parametrize("case", TAIL_CASES, ids=[c[0] for c in TAIL_CASES])
def test_fused_tail_conv_matches_oracle(hip_lib, case, nchw):
    """Wt . relu(bn(conv(x))) + bt in one launch; the oracle rounds the intermediate to fp16 as the unfused path does."""
    name, N, Cin, H, W, Cout, k, stride, pad, transposed, nt = case
    dev, dtype = torch.device("cuda:0"), torch.float16
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    fan = (Cin * 4) if transposed else Cin * k * k
    w = synth.normal(13, name + ".w", wshape, std=(2.0 / fan) ** 0.5).half().float()
    x = synth.normal(13, name + ".x", (N, Cin, H, W)).half().float()
    bn = {"weight": synth.uniform(13, name + ".g", (Cout,), 0.5, 1.5), "bias": synth.normal(13, name + ".be", (Cout,), 0.1),
          "running_mean": synth.normal(13, name + ".m", (Cout,), 0.1), "running_var": synth.uniform(13, name + ".v", (Cout,), 0.5, 1.5),
          "eps": 1e-5}
    wt = synth.normal(13, name + ".wt", (nt, Cout, 1, 1), std=(1.0 / Cout) ** 0.5).half().float()
    bt = synth.normal(13, name + ".bt", (nt,), 0.2)
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
        run_program(prog)
        got = y.cpu()
    else:
        ybuf = torch.full((N, Ho, Wo, 40), 7.0, dtype=dtype, device=dev)
        yv = ActView(ybuf, nt, 4)
        layer.record(prog, xv, yv)
        run_program(prog)
        got = view_to_nchw(yv)
        assert torch.all(ybuf[..., :4] == 7.0) and torch.all(ybuf[..., 4 + nt:] == 7.0), "wrote outside its channel slice"
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert err <= (2e-3 if nchw else 1e-2) * scale, f"{name}: max abs err {err:.3e} (scale {scale:.2f})"


# ---- regression: workgroup recycling (thousands of tiles) must not change a single bit ------------------------
@pytest.mark.parametrize("case", [("stem_pose", 96, 3, 128, 96, 64, 7, 2, 3, False), ("stem_flow", 24, 6, 192, 256, 64, 7, 2, 3, False),
                                  ("halo_3x3", 96, 64, 32, 24, 64, 3, 1, 1, False), ("halo_deconv", 48, 256, 16, 12, 256, 4, 2, 1, True)],
                         ids=lambda c: c[0])
def test_patch_kernels_are_deterministic_when_workgroups_are_recycled(hip_lib, case):
    """The LDS-patch kernels once read patch rows before they had landed — only when a CU ran several workgroups one
    after the other (> ~1500 tiles), ~0.1 % of the tiles.  Every variant: 4 runs bit-identical and equal to the plain
    direct-to-LDS variant within fp16 rounding."""
    import ctypes
    from flowtrack.pytorch_amd.hip_ops import new_act, new_rowpacked_act
    name, N, Cin, H, W, Cout, k, stride, pad, transposed = case
    dev, dtype = torch.device("cuda:0"), torch.float16
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    w = synth.normal(21, name + ".w", wshape, std=(2.0 / (Cin * k * k)) ** 0.5)
    layer = FusedConv(w, dtype=dtype, device=dev, stride=stride, pad=pad, transposed=transposed, act="relu", label=name)
    if Cin <= 16:
        xv = new_rowpacked_act(N, H, W, Cin, pad, dtype, dev)
        xv.t[:, :, pad:pad + W, :Cin] = synth.normal(21, name + ".x", (N, H, W, Cin)).to(device=dev, dtype=dtype)
    else:
        xv = new_act(N, H, W, Cin, dtype, dev)
        xv.t[..., :Cin] = synth.normal(21, name + ".x", (N, H, W, Cin)).to(device=dev, dtype=dtype)
    Ho, Wo = layer.out_hw(H, W)
    yv = new_act(N, Ho, Wo, Cout, dtype, dev)
    prog = make_program()
    layer.record(prog, xv, yv)
    d = prog.conv_records[0][3]
    hints = (ctypes.c_int * 64)()
    n = hip_lib.ft_conv_tile_candidates(ctypes.byref(d), hints, 64)
    halo = [int(h) for h in hints[:n] if (int(h) >> 30) & 1]
    assert halo, "no LDS-patch variant offered"
    d.tile_hint = next(int(h) for h in hints[:n] if not (int(h) >> 30) & 1)
    run_program(prog)
    ref = yv.t.float().clone()
    scale = max(1.0, ref.abs().max().item())
    for h in halo:
        d.tile_hint = h
        outs = []
        for _ in range(4):
            yv.t.fill_(3.0)
            run_program(prog)
            outs.append(yv.t.clone())
        assert all(torch.equal(outs[0], o) for o in outs[1:]), f"{name} hint {h:#x}: runs differ"
        assert (outs[0].float() - ref).abs().max().item() <= 2e-2 * scale, f"{name} hint {h:#x}: differs from the plain variant"


# ---- ResNet stem with the max-pool fused (ft_conv_desc.pool): conv1 -> bn1 -> relu -> maxpool, resnet.py:19-23 ----------
STEM_POOL_CASES = [("r50_crop", 2, 256, 192), ("small_ragged_x", 2, 64, 48), ("ragged_xy", 3, 40, 56), ("recycle", 40, 256, 192),
                   ("ragged_big", 24, 200, 168)]      # 1008 tiles, ragged both ways


@pytest.mark.parametrize("case", STEM_POOL_CASES, ids=[c[0] for c in STEM_POOL_CASES])
def test_stem_with_fused_maxpool_matches_oracle_and_separate_launches(hip_lib, case):
    from flowtrack.pytorch_amd.hip_ops import new_act, new_rowpacked_act, record_maxpool, record_pack_input
    name, N, H, W = case
    dev, dtype, seed = torch.device("cuda:0"), torch.float16, 31
    w = synth.normal(seed, name + ".w", (64, 3, 7, 7), std=(2.0 / 147) ** 0.5)
    bn = {"weight": synth.uniform(seed, name + "g", (64,), 0.5, 1.5), "bias": synth.normal(seed, name + "b", (64,), 0.3),
          "running_mean": synth.normal(seed, name + "m", (64,), 0.1), "running_var": synth.uniform(seed, name + "v", (64,), 0.5, 1.5), "eps": 1e-5}
    x = synth.normal(seed, name + ".x", (N, 3, H, W))
    layer = FusedConv(w, stride=2, pad=3, bn=bn, act="relu", dtype=dtype, device=dev, label=name)
    xs = x.to(dev)
    Hp, Wp = H // 4, W // 4

    fused_in = new_rowpacked_act(N, H, W, 3, 5, dtype, dev)
    pooled = new_act(N, Hp, Wp, 64, dtype, dev)
    pooled.t.fill_(9.0)
    prog = make_program()
    record_pack_input(prog, xs, fused_in)
    layer.record(prog, fused_in, pooled, pool=True)
    run_program(prog)
    got = view_to_nchw(pooled)

    sep_in = new_rowpacked_act(N, H, W, 3, 3, dtype, dev)
    a1 = new_act(N, H // 2, W // 2, 64, dtype, dev)
    ref_pooled = new_act(N, Hp, Wp, 64, dtype, dev)
    prog2 = make_program()
    record_pack_input(prog2, xs, sep_in)
    layer.record(prog2, sep_in, a1)
    record_maxpool(prog2, a1, ref_pooled)
    run_program(prog2)
    assert torch.equal(got, view_to_nchw(ref_pooled)), f"{name}: fused stem + pool differs from conv + maxpool launches"

    xh = x.half().float()
    want = F.max_pool2d(_reference(xh, w, None, bn, 2, 3, False, "relu", None), 3, 2, 1)
    scale = max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() <= 2e-2 * scale
    pooled.t.fill_(7.0)
    run_program(prog)
    assert torch.equal(view_to_nchw(pooled), got)

    # ft_conv_desc.x_nchw_f32: the same launch gathering its patches from the NCHW fp32 input itself (no pack launch, the packed
    # view is geometry only): bit-identical
    planar = new_act(N, Hp, Wp, 64, dtype, dev)
    planar.t.fill_(5.0)
    geom = new_rowpacked_act(N, H, W, 3, 5, dtype, "meta")
    prog3 = make_program()
    layer.record(prog3, geom, planar, pool=True, x_nchw=xs)
    assert [c[0] for c in prog3.calls if not c[0].startswith("__")] in (["ft_conv2d_fwd_ws"], ["ft_conv2d_fwd"])
    run_program(prog3)
    assert torch.equal(view_to_nchw(planar), got), f"{name}: the stem on the NCHW input differs from pack + stem"
    with pytest.raises(FlowtrackHipError):
        layer.record(make_program(), geom, planar, pool=True)                        # geometry-only view without the input
    with pytest.raises(FlowtrackHipError):
        layer.record(make_program(), fused_in, planar, pool=True, x_nchw=xs[:, :2])  # wrong plane count


@pytest.mark.parametrize("shape", [(2, 3, 16, 24, 5), (1, 3, 9, 18, 3), (3, 2, 8, 32, 3), (2, 3, 6, 20, 3), (2, 3, 256, 192, 5)], ids=str)
def test_pack_input_rowpacked_layout(hip_lib, shape):
    """ft_pack_nchw_to_nhwc into the row-packed stem layout: data in columns [lpad, lpad + W), every pad column and the
    padding channel zero even when the buffer held garbage (fast 4-pixel path when W % 4 == 0, generic path otherwise)."""
    from flowtrack.pytorch_amd.hip_ops import new_rowpacked_act, record_pack_input
    N, C, H, W, pad = shape
    dev = torch.device("cuda:0")
    x = synth.normal(5, "pack" + str(shape), (N, C, H, W)).to(dev)
    v = new_rowpacked_act(N, H, W, C, pad, torch.float16, dev)
    v.t.fill_(9.0)
    prog = make_program()
    record_pack_input(prog, x, v)
    run_program(prog)
    want = torch.zeros_like(v.t)
    want[:, :, pad:pad + W, :C] = x.permute(0, 2, 3, 1).half()
    assert torch.equal(v.t, want)


@pytest.mark.parametrize("act", [None, "relu", "leaky"])
def test_epilogue_keeps_non_finite_values(hip_lib, act):
    """conv_common.h: apply_act on NaN / +-inf: a NaN accumulator stays NaN through every activation
    (torch's F.relu / F.leaky_relu do the same), +inf stays +inf, -inf follows torch — including ReLU(-inf) = 0, which the
    two-instruction max(v, 0 * v) of rounds 2-5 left at -inf."""
    dev = torch.device("cuda:0")
    N, C, H, W = 1, 32, 4, 8
    w = torch.eye(C).reshape(C, C, 1, 1)
    x = synth.normal(3, "nf.x", (N, C, H, W))
    x[0, 1, 0, 0] = float("nan")
    x[0, 2, 1, 1] = float("inf")
    x[0, 3, 2, 2] = float("-inf")
    layer = FusedConv(w, dtype=torch.float32, device=dev, act=act, slope=0.1, label="nonfinite")
    xv = nchw_to_view(x, torch.float32, dev)
    yv = ActView(torch.zeros((N, H, W, C), dtype=torch.float32, device=dev), C, 0)
    prog = make_program()
    layer.record(prog, xv, yv)
    run_program(prog)
    got = view_to_nchw(yv)
    # an identity 1x1 conv: every output pixel whose 32 inputs are finite is act(x); the three poisoned PIXELS are non-finite in
    # all channels of the GEMM row that touched them (0 * NaN = NaN, 0 * inf = NaN) — what torch's conv gives as well
    want = _reference(x, w, None, None, 1, 0, False, act, None)
    fin = torch.isfinite(want)
    if act == "relu":
        assert got[0, 3, 2, 2] == 0 and want[0, 3, 2, 2] == 0
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert torch.allclose(got[fin], want[fin], atol=1e-6)
    assert torch.isnan(got[0, 1, 0, 0]) and torch.isnan(got[0, 0, 0, 0])          # NaN in -> NaN out, under every activation
    assert got[0, 2, 1, 1] == float("inf")
    if act != "relu":
        assert got[0, 3, 2, 2] == (float("-inf") if act is None else float("-inf") * 0.1)
