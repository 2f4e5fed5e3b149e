"""FlowNet2 operators and networks on the GPU (through the C ABI) vs the CPU oracle and the goldens
written by the imported reference.  Tolerances: fp32 ops 1e-4..1e-3 max-abs (north_star: 1e-3)."""
import ctypes
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from flowtrack.pytorch_amd import _lib, synth
from flowtrack.pytorch_amd._lib import check
from flowtrack.pytorch_amd.flownet import models
from oracle import flow_ref, ops_ref

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "flow_golden.npz"))
SEED = int(G["seed"])
ARGS = types.SimpleNamespace(rgb_max=255.0, fp16=False)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


# ---- operators ------------------------------------------------------------------------------------
CORR_CASES = [
    # B, C, H, W, pad, k, max_disp, s1, s2
    (2, 16, 12, 14, 4, 1, 4, 1, 2),      # FlowNetC-like, small
    (1, 256, 12, 16, 20, 1, 20, 1, 2),   # FlowNetC parameters (FlowNetC.py:31), reduced spatial size
    (1, 8, 10, 9, 3, 3, 2, 1, 1),        # kernel_size 3
    (1, 8, 16, 15, 4, 1, 4, 2, 2),       # stride1 2
    (1, 5, 9, 11, 2, 1, 3, 1, 1),        # pad < max_disp: output smaller than input
]


@pytest.mark.parametrize("case", CORR_CASES, ids=[str(c) for c in CORR_CASES])
def test_correlation_nchw(hip_lib, oracle_lib, case):
    B, C, H, W, pad, k, d, s1, s2 = case
    a = synth.normal(3, "corr_a", (B, C, H, W)).numpy()
    b = synth.normal(3, "corr_b", (B, C, H, W)).numpy()
    want = ops_ref.correlation_c(a, b, pad, k, d, s1, s2)
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(hip_lib.ft_correlation_out_shape(C, H, W, pad, k, d, s1, s2, ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow)))
    assert (B, oc.value, oh.value, ow.value) == want.shape
    ga, gb = _cuda(a), _cuda(b)
    out = torch.full(want.shape, 9.0, dtype=torch.float32, device="cuda")
    check(hip_lib.ft_correlation_fwd(ga.data_ptr(), gb.data_ptr(), out.data_ptr(), B, C, H, W, pad, k, d, s1, s2, 1, _stream()))
    torch.cuda.synchronize()
    err = np.abs(out.cpu().numpy() - want).max()
    assert err <= 1e-5, f"max abs err {err:.3e}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["fp32", "fp16"])
@pytest.mark.parametrize("shape", [(1, 256, 12, 16), (2, 64, 9, 37), (2, 256, 24, 64), (1, 256, 5, 83), (2, 256, 7, 40), (3, 256, 48, 64)],
                         ids=["c256", "ragged", "c256_w64", "c256_w83", "c256_w40_hodd", "c256_flownetc_map"])
def test_correlation_nhwc_fused(hip_lib, oracle_lib, dtype, shape):
    """in-network form: NHWC features -> LeakyReLU'd cost volume in a channel slice of the concat buffer."""
    B, C, H, W = shape
    a = synth.normal(4, "corr_a", shape)
    b = synth.normal(4, "corr_b", shape)
    if dtype == torch.float16:
        a, b = a.half().float(), b.half().float()
    want = ops_ref.correlation_c(a.numpy(), b.numpy(), 20, 1, 20, 1, 2)
    want = np.where(want > 0, want, 0.1 * want)
    fa = a.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)
    fb = b.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)
    y = torch.full((B, H, W, 480), 5.0, dtype=dtype, device="cuda")
    check(hip_lib.ft_correlation_nhwc_fwd(fa.data_ptr(), fb.data_ptr(), y.data_ptr(), B, C, H, W, 20, 2, C, 480, 32,
                                          _lib.FT_ACT_LEAKY, 0.1, _lib.dtype_code(dtype), _stream()))
    torch.cuda.synchronize()
    got = y[..., 32:32 + 441].permute(0, 3, 1, 2).float().cpu().numpy()
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert np.abs(got - want).max() <= tol
    assert torch.all(y[..., :32] == 5.0) and torch.all(y[..., 473:] == 5.0)


def test_correlation_rows_kernel_is_deterministic_under_memory_load(hip_lib):
    """The matrix-core correlation orders its look-ahead ring rows and band stores by hand-counted vmcnt waits alone (round 6: the
    compiler's hidden per-step queue drains are gone): 60 launches at FlowNetC's shape [16,256,48,64] and a ragged one, the second
    half with another stream streaming through HBM, must all equal the first bit for bit (tools/dev/corr_stress.py: the long form)."""
    for (B, H, W) in ((16, 48, 64), (3, 47, 61)):
        f = (synth.normal(6, f"corr_det{H}", (2 * B, H, W, 256)) * 1.0).to("cuda", torch.float16)
        y = torch.zeros((B, H, W, 480), dtype=torch.float16, device="cuda")

        def run():
            check(hip_lib.ft_correlation_nhwc_fwd(f[:B].data_ptr(), f[B:].data_ptr(), y.data_ptr(), B, 256, H, W, 20, 2, 256, 480, 32,
                                                  _lib.FT_ACT_LEAKY, 0.1, _lib.FT_F16, _stream()))
        run()
        torch.cuda.synchronize()
        ref = y[..., 32:473].clone()
        side, junk = torch.cuda.Stream(), torch.empty((128 << 20,), dtype=torch.uint8, device="cuda")
        for it in range(60):
            if it >= 30:
                with torch.cuda.stream(side):
                    junk.add_(1)
            y.fill_(3.0)
            run()
            torch.cuda.synchronize()
            assert torch.equal(y[..., 32:473], ref), (B, H, W, it)


def test_correlation_impulse_known_answers_gpu(hip_lib):
    """The hand-derived impulse responses (tests/golden/correlation_kat.npz) through the C ABI: the reference-API kernel on
    every case, the in-network NHWC kernels (fp32 VALU form and the fp16 matrix-core forms) where their fixed parameters
    (kernel 1, stride1 1, pad = max_displacement) apply.  A transposed displacement grid or a flipped sign fails here."""
    import os
    from conftest import GOLDEN
    G = np.load(os.path.join(GOLDEN, "correlation_kat.npz"))
    for n in sorted({k.split(".")[0] for k in G.files}):
        pad, k, d, s1, s2 = (int(v) for v in G[n + ".params"])
        a, b, want = G[n + ".in1"], G[n + ".in2"], G[n + ".out"]
        B, C, H, W = a.shape
        ga, gb = _cuda(a), _cuda(b)
        out = torch.full(want.shape, 9.0, dtype=torch.float32, device="cuda")
        check(hip_lib.ft_correlation_fwd(ga.data_ptr(), gb.data_ptr(), out.data_ptr(), B, C, H, W, pad, k, d, s1, s2, 1, _stream()))
        torch.cuda.synchronize()
        assert np.abs(out.cpu().numpy() - want).max() <= 1e-6, n
        if k == 1 and s1 == 1 and pad == d:
            D2 = want.shape[1]
            for dtype, cpad in ((torch.float32, 8), (torch.float16, 8), (torch.float16, 256)):   # 256 channels: the MFMA kernels
                if cpad == 256 and not (s2 == 2 and d % 2 == 0):
                    continue
                fa = torch.zeros((B, H, W, cpad), dtype=dtype, device="cuda")
                fb = torch.zeros((B, H, W, cpad), dtype=dtype, device="cuda")
                fa[..., :C] = torch.from_numpy(a).permute(0, 2, 3, 1).to("cuda", dtype)
                fb[..., :C] = torch.from_numpy(b).permute(0, 2, 3, 1).to("cuda", dtype)
                y = torch.full((B, H, W, D2 + 7), 5.0, dtype=dtype, device="cuda")
                check(hip_lib.ft_correlation_nhwc_fwd(fa.data_ptr(), fb.data_ptr(), y.data_ptr(), B, cpad, H, W, d, s2, cpad, D2 + 7, 0,
                                                      _lib.FT_ACT_NONE, 0.0, _lib.dtype_code(dtype), _stream()))
                torch.cuda.synchronize()
                got = y[..., :D2].permute(0, 3, 1, 2).float().cpu().numpy() * (cpad / C)      # nelems = the padded channel count
                assert np.abs(got - want).max() <= 1e-3, (n, dtype, cpad)


def test_resample2d_and_channelnorm(hip_lib, oracle_lib):
    B, C, H, W = 2, 3, 24, 40
    img = synth.normal(5, "img", (B, C, H, W)).numpy()
    flow = synth.flow_field(5, B, H, W, magnitude=6.0).numpy()
    flow[0, :, 0, 0] = (-100.0, 250.0)  # far out of frame: border clamp, weights NOT renormalised
    flow[1, :, 3, 3] = (0.0, 0.0)
    want = ops_ref.resample2d_c(img, flow)
    out = torch.empty((B, C, H, W), dtype=torch.float32, device="cuda")
    gimg, gflow = _cuda(img), _cuda(flow)   # keep the device buffers alive across the async launches
    check(hip_lib.ft_resample2d_fwd(gimg.data_ptr(), gflow.data_ptr(), out.data_ptr(), B, C, H, W, _stream()))
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-5
    # zero flow is the identity; integer flow is a shift
    zero = torch.zeros((B, 2, H, W), device="cuda")
    check(hip_lib.ft_resample2d_fwd(gimg.data_ptr(), zero.data_ptr(), out.data_ptr(), B, C, H, W, _stream()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), img)
    nrm = torch.empty((B, 1, H, W), dtype=torch.float32, device="cuda")
    check(hip_lib.ft_channelnorm_fwd(gimg.data_ptr(), nrm.data_ptr(), B, C, H, W, _stream()))
    torch.cuda.synchronize()
    assert np.abs(nrm.cpu().numpy() - ops_ref.channelnorm_c(img)).max() <= 1e-6
    # both ChannelNorm kernels: HW % 4 == 0 takes the 16-byte form (C = 3 with all loads in flight, other C in a loop), anything
    # else (odd plane size, a misaligned plane pointer) the one-pixel form; same per-pixel arithmetic: identical bits
    for (b2, c2, h2, w2, off) in ((2, 3, 12, 20, 0), (1, 5, 8, 12, 0), (2, 3, 7, 9, 0), (1, 3, 8, 8, 1), (3, 1, 4, 4, 0)):
        im = synth.normal(8, f"cn{c2}{h2}{w2}", (b2 * c2 * h2 * w2 + 4,))
        g = im.cuda()
        o = torch.full((b2 * h2 * w2 + 4,), -1.0, device="cuda")
        check(hip_lib.ft_channelnorm_fwd(g.data_ptr() + 4 * off, o.data_ptr() + 4 * off, b2, c2, h2, w2, _stream()))
        torch.cuda.synchronize()
        src = im[off:off + b2 * c2 * h2 * w2].reshape(b2, c2, h2, w2).numpy()
        want_n = ops_ref.channelnorm_c(src).reshape(-1)
        got_n = o[off:off + b2 * h2 * w2].cpu().numpy()
        assert np.abs(got_n - want_n).max() <= 1e-6, (b2, c2, h2, w2, off)
        assert float(o[off + b2 * h2 * w2:].max()) == -1.0 and (off == 0 or float(o[0]) == -1.0)
    # the other kernel forms: a width that is no multiple of 4 and more than 4 channels (one pixel per thread), 1 / 2 / 4
    # channels (four pixels per thread); out-of-frame flows in every case
    for (b2, c2, h2, w2) in ((1, 3, 10, 13), (1, 5, 8, 12), (2, 1, 9, 16), (1, 2, 7, 8), (1, 4, 6, 20)):
        img2 = synth.normal(6, f"img{c2}{w2}", (b2, c2, h2, w2)).numpy()
        flow2 = synth.flow_field(6, b2, h2, w2, magnitude=5.0).numpy()
        flow2[0, :, h2 - 1, w2 - 1] = (300.0, -40.0)
        out2 = torch.empty((b2, c2, h2, w2), dtype=torch.float32, device="cuda")
        gi, gf = _cuda(img2), _cuda(flow2)
        check(hip_lib.ft_resample2d_fwd(gi.data_ptr(), gf.data_ptr(), out2.data_ptr(), b2, c2, h2, w2, _stream()))
        torch.cuda.synchronize()
        assert np.abs(out2.cpu().numpy() - ops_ref.resample2d_c(img2, flow2)).max() <= 1e-5, (b2, c2, h2, w2)


def test_resample2d_window_kernel_paths(hip_lib, oracle_lib):
    """The LDS-window form of Resample2d (16 x 64 tiles, csrc/flow_ops.hip): ragged maps with several tiles in both directions;
    incoherent per-pixel flows (sigma 4 px: clipped windows + gathers for the pixels outside), one image whose flow spreads over the
    whole map (no window: every pixel gathers), far out-of-frame flows (border clamp without renormalisation,
    Resample2d_kernel.cu:42-59), 1..4 channels."""
    for (B, C, H, W) in ((3, 3, 70, 150), (2, 4, 33, 200), (2, 1, 16, 64), (1, 2, 50, 65), (3, 3, 96, 256)):
        img = synth.normal(8, f"img{C}{W}", (B, C, H, W)).numpy()
        flow = (synth.normal(8, f"flow{C}{W}", (B, 2, H, W)) * 4.0).numpy()
        # round 6: a box that does not fit LDS keeps a window CLIPPED around the tile, pixels outside it gather: sigma 4 px at the
        # default budget is that case (both piece sizes: W % 4 == 0 and not); a sprinkle of 10-25 px vectors lands outside the clip
        far = synth.uniform(10, f"far{C}{W}", (B, 1, H, W), 0.0, 1.0).numpy() < 0.03
        flow = np.where(far, flow * 5.0, flow).astype(np.float32)
        flow[B - 1] = (synth.normal(9, f"wide{C}{W}", (2, H, W)) * 60.0).numpy()       # spread >> four windows: no window, every pixel gathers
        flow[0, :, 0, 0] = (-1000.0, 2500.0)
        flow[0, :, H - 1, W - 1] = (1e9, -1e9)
        want = ops_ref.resample2d_c(img, flow)
        out = torch.full((B, C, H, W), 7.0, dtype=torch.float32, device="cuda")
        gi, gf = _cuda(img), _cuda(flow)
        check(hip_lib.ft_resample2d_fwd(gi.data_ptr(), gf.data_ptr(), out.data_ptr(), B, C, H, W, _stream()))
        torch.cuda.synchronize()
        assert np.abs(out.cpu().numpy() - want).max() <= 1e-5, (B, C, H, W)


def test_fused_mean_pack_pair_matches_the_two_launches(hip_lib):
    """ft_flow_mean_pack_pair (rgb mean + (x - mean) / rgb_max + row-packing in ONE launch, the workgroups of a sample exchanging
    their partial sums as tagged 8-byte words inside the launch; models.py:255-257) against ft_flow_rgb_mean + ft_flow_pack_pair:
    the same mean up to the summation order, hence the same packed view up to that (fp32) and to the last fp16 bit almost
    everywhere; pad columns zero; REPEATED launches on one state buffer with changing inputs (the epoch logic: a stale tag of the
    previous launch must never satisfy the sweep); one / two / three / thirty-two workgroups per sample; ragged last workgroup."""
    from flowtrack.pytorch_amd.hip_ops import new_rowpacked_act
    for (B, H, W, pad) in ((16, 384, 512, 3), (3, 130, 100, 3), (2, 64, 64, 1), (5, 50, 260, 3)):
        words = int(hip_lib.ft_flow_mean_pack_pair_state_words(B, H, W))
        assert words > 0
        state = torch.zeros(words, dtype=torch.int64, device="cuda")
        for dtype, tol in ((torch.float16, 1e-3), (torch.float32, 1e-6)):
            for rep in range(3):
                pair = (synth.frame_pairs(40 + rep, B, H, W) * (1.0 + 0.5 * rep)).cuda()
                partial = torch.empty(B * 3 * _lib.FT_RGB_MEAN_SPLITS, device="cuda")
                mean2 = torch.empty(B * 3, device="cuda")
                want = new_rowpacked_act(B, H, W, 6, pad, dtype, "cuda")
                want.t.fill_(5.0)
                check(hip_lib.ft_flow_rgb_mean(pair.data_ptr(), B, H, W, partial.data_ptr(), mean2.data_ptr(), _stream()))
                check(hip_lib.ft_flow_pack_pair(pair.data_ptr(), mean2.data_ptr(), ctypes.c_float(255.0), want.t.data_ptr(), B, H, W, 0,
                                                want.lpad, want.wpitch, _lib.dtype_code(dtype), _stream()))
                want3 = new_rowpacked_act(2 * B, H, W, 3, pad, dtype, "cuda")
                want3.t.fill_(5.0)
                check(hip_lib.ft_flow_pack_pair(pair.data_ptr(), mean2.data_ptr(), ctypes.c_float(255.0), want3.t.data_ptr(), B, H, W, 1,
                                                want3.lpad, want3.wpitch, _lib.dtype_code(dtype), _stream()))
                got = new_rowpacked_act(B, H, W, 6, pad, dtype, "cuda")
                got.t.fill_(7.0)
                got3 = new_rowpacked_act(2 * B, H, W, 3, pad, dtype, "cuda")
                got3.t.fill_(7.0)
                mean1 = torch.empty(B * 3, device="cuda")
                # rep 0: both views; rep 1: the 6-channel view alone; rep 2: the siamese view alone
                p6 = got.t.data_ptr() if rep != 2 else None
                p3 = got3.t.data_ptr() if rep != 1 else None
                check(hip_lib.ft_flow_mean_pack_pair(pair.data_ptr(), ctypes.c_float(255.0), p6, got.lpad, got.wpitch, p3, got3.lpad,
                                                     got3.wpitch, B, H, W, _lib.dtype_code(dtype), state.data_ptr(), mean1.data_ptr(),
                                                     _stream()))
                torch.cuda.synchronize()
                assert int(state[-1]) == 0, "a workgroup timed out waiting for its sample's partial sums"
                ref = pair.view(B, 3, -1).double().mean(-1).flatten().float()
                assert (mean1 - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() and (mean1 - mean2).abs().max().item() <= 1e-3
                if p3 is not None:
                    d3 = (got3.t.float() - want3.t.float()).abs()
                    assert d3.max().item() <= tol, (B, H, W, dtype, rep, d3.max().item())
                    assert torch.all(got3.t[:, :, :got3.lpad] == 0) and torch.all(got3.t[:, :, got3.lpad + W:] == 0) and torch.all(got3.t[..., 3:] == 0)
                if p6 is None:
                    continue
                d = (got.t.float() - want.t.float()).abs()
                assert d.max().item() <= tol, (B, H, W, dtype, rep, d.max().item())
                assert torch.all(got.t[:, :, :got.lpad] == 0) and torch.all(got.t[:, :, got.lpad + W:] == 0) and torch.all(got.t[..., 6:] == 0)
    assert int(hip_lib.ft_flow_mean_pack_pair_state_words(2, 64, 62)) == 0          # W % 4 != 0: the two launches
    assert int(hip_lib.ft_flow_mean_pack_pair_state_words(1, 2048, 2048)) == 0      # more than 42 workgroups per sample


def test_mean_fold_pack_sums_and_fold_tables(hip_lib):
    """ft_flow_pack_pair_sums + ft_flow_mean_fold (the rgb mean of models.py:255-257 folded into conv1): the padded view holds
    fp16(x / rgb_max) inside and fp16(mean / rgb_max) in every padding pixel, the mean matches the fp64 one, the per-sample
    shift is shift - scale * sum_c m16[c] * wsum[co][c]; ragged last row chunk, pitch wider than W + 2 pad."""
    for (B, H, W, pad, extra) in ((3, 50, 72, 3, 0), (2, 64, 64, 3, 2), (16, 384, 512, 3, 0)):
        pair = (synth.frame_pairs(77, B, H, W)).cuda()
        wpitch = W + 2 * pad + extra
        y = torch.full((B, H + 2 * pad, wpitch, 8), 9.0, dtype=torch.float16, device="cuda")
        nchunk = int(hip_lib.ft_flow_pack_pair_sums_chunks(H))
        assert nchunk == (H + 3) // 4
        partial = torch.empty(B * 3 * nchunk, device="cuda")
        check(hip_lib.ft_flow_pack_pair_sums(pair.data_ptr(), ctypes.c_float(255.0), y.data_ptr(), B, H, W, pad, wpitch, _lib.FT_F16,
                                             partial.data_ptr(), _stream()))
        g = torch.Generator().manual_seed(5)
        wsum = torch.randn((64, 8), generator=g)
        scale = torch.rand(64, generator=g) + 0.5
        shift = torch.randn(64, generator=g)
        wsum_g, scale_g, shift_g = wsum.cuda(), scale.cuda(), shift.cuda()
        for use_scale in (True, False):
            shn = torch.empty((B, 64), device="cuda")
            mean = torch.empty(B * 3, device="cuda")
            check(hip_lib.ft_flow_mean_fold(partial.data_ptr(), ctypes.c_float(255.0), y.data_ptr(), B, H, W, pad, wpitch, _lib.FT_F16,
                                            wsum_g.data_ptr(), scale_g.data_ptr() if use_scale else None,
                                            shift_g.data_ptr(), 64, shn.data_ptr(), mean.data_ptr(), _stream()))
            torch.cuda.synchronize()
            ref = pair.view(B, 3, -1).double().mean(-1)
            assert (mean.view(B, 3).double() - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
            m16 = (mean.view(B, 3) / 255.0).half()
            got = y.cpu()
            inner = got[:, pad:pad + H, pad:pad + W]
            want = torch.cat((pair[:, :, 0], pair[:, :, 1]), 1).permute(0, 2, 3, 1).cpu()
            assert torch.equal(inner[..., :6], (want / 255.0).half()) and torch.all(inner[..., 6:] == 0)
            mask = torch.ones((H + 2 * pad, wpitch), dtype=torch.bool)
            mask[pad:pad + H, pad:pad + W] = False
            padpx = got[:, mask]                                      # [B, npad, 8]
            want_pad = torch.cat((m16, m16, torch.zeros((B, 2), dtype=torch.float16, device="cuda")), 1).cpu()
            assert torch.equal(padpx, want_pad[:, None, :].expand_as(padpx))
            mf = m16.float().cpu().double()
            corr = mf @ wsum[:, 0:3].double().t() + mf @ wsum[:, 3:6].double().t()            # [B, 64]
            want_shn = shift.double()[None] - (scale.double()[None] if use_scale else 1.0) * corr
            assert (shn.cpu().double() - want_shn).abs().max().item() <= 1e-5
    # refused: W % 4 != 0, fp32
    y = torch.empty((1, 14, 16, 8), dtype=torch.float16, device="cuda")
    pair = synth.frame_pairs(1, 1, 8, 10).cuda()
    partial = torch.empty(6, device="cuda")
    assert hip_lib.ft_flow_pack_pair_sums(pair.data_ptr(), ctypes.c_float(255.0), y.data_ptr(), 1, 8, 10, 3, 16, _lib.FT_F16,
                                          partial.data_ptr(), _stream()) == _lib.FT_ERR_UNSUPPORTED
    assert hip_lib.ft_flow_pack_pair_sums(pair.data_ptr(), ctypes.c_float(255.0), y.data_ptr(), 1, 8, 8, 3, 16, _lib.FT_F32,
                                          partial.data_ptr(), _stream()) == _lib.FT_ERR_UNSUPPORTED


def test_mean_fold_flownet2s_matches_the_two_launch_path(hip_lib, monkeypatch):
    """FlowNet2S fp16 with the rgb mean folded into conv1 (default where conv1 runs on the persistent stem) against the same
    network on ft_flow_rgb_mean + ft_flow_pack_pair, both against the fp32 HIP path: the fold's plan really carries the fold
    launches and no mean launch, the two fp16 flows agree to fp16 noise, neither is further from fp32 than the other by more
    than that noise; also with BatchNorm (scale != 1 in the per-sample shift) and with a constant image (x - mean = 0: the
    fold's cancellation case)."""
    B, H, W = 4, 256, 256
    for bn in (False, True):
        pair = synth.frame_pairs(SEED + 9, B, H, W).cuda()
        m32, _ = _build(models.FlowNet2S, SEED + 4, torch.float32, batchNorm=bn)
        want = m32(pair).cpu()
        monkeypatch.setattr(models, "MEAN_FOLD", True)
        mf, _ = _build(models.FlowNet2S, SEED + 4, torch.float16, batchNorm=bn)
        got_fold = mf(pair).cpu()
        names = [c[0] for c in mf.plan_for(B, H, W).prog.calls]
        assert "ft_flow_mean_fold" in names and "ft_flow_pack_pair_sums" in names and "ft_flow_rgb_mean" not in names
        monkeypatch.setattr(models, "MEAN_FOLD", False)
        m2, _ = _build(models.FlowNet2S, SEED + 4, torch.float16, batchNorm=bn)
        got_two = m2(pair).cpu()
        names = [c[0] for c in m2.plan_for(B, H, W).prog.calls]
        assert "ft_flow_mean_fold" not in names and "ft_flow_rgb_mean" in names
        mag = torch.norm(want, dim=1).mean().item()
        e_fold, e_two, e_between = flow_ref.epe(got_fold, want), flow_ref.epe(got_two, want), flow_ref.epe(got_fold, got_two)
        print("bn", bn, "EPE fold", e_fold, "two-launch", e_two, "between", e_between, "mean |flow|", mag)
        bar = 0.02 * max(mag, 1.0) + 0.05
        assert e_fold <= bar and e_two <= bar and e_between <= bar
        assert e_fold <= 1.5 * e_two + 0.01
        # graph replays are bit-identical
        assert torch.equal(mf(pair).cpu(), got_fold)
        # constant frames: the centred input is exactly zero in the reference
        monkeypatch.setattr(models, "MEAN_FOLD", True)
        flat = torch.full((B, 3, 2, H, W), 200.0, device="cuda")
        f_fold, f_32 = mf(flat).cpu(), m32(flat).cpu()
        assert flow_ref.epe(f_fold, f_32) <= bar


def test_upsample_and_normalise(hip_lib, oracle_lib):
    x = synth.normal(6, "flow2", (2, 2, 6, 9)).numpy()
    y = torch.empty((2, 2, 24, 36), dtype=torch.float32, device="cuda")
    gx = _cuda(x)
    check(hip_lib.ft_upsample_bilinear4x(gx.data_ptr(), y.data_ptr(), 2, 2, 6, 9, 20.0, _stream()))
    torch.cuda.synchronize()
    want = torch.nn.functional.interpolate(torch.from_numpy(x) * 20.0, scale_factor=4, mode="bilinear", align_corners=False).numpy()
    assert np.abs(y.cpu().numpy() - want).max() <= 1e-5
    assert np.abs(y.cpu().numpy() - ops_ref.upsample4x_c(x, 20.0)).max() <= 1e-5
    # rgb mean + (x - mean) / rgb_max, both packings; W % 4 == 0 takes the 4-pixels-per-thread kernels, W = 62 the scalar ones
    for W in (64, 62):
        B, H = 2, 64
        pair = synth.frame_pairs(6, B, H, W)
        gp = pair.cuda()
        partial = torch.empty(B * 3 * _lib.FT_RGB_MEAN_SPLITS, device="cuda")
        mean = torch.empty(B * 3, device="cuda")
        check(hip_lib.ft_flow_rgb_mean(gp.data_ptr(), B, H, W, partial.data_ptr(), mean.data_ptr(), _stream()))
        want_mean = pair.view(B, 3, -1).mean(-1)
        torch.cuda.synchronize()
        assert (mean.cpu().view(B, 3) - want_mean).abs().max() <= 1e-3
        xn = flow_ref._normalise(pair, 255.0)
        for dtype, code, tol in ((torch.float32, _lib.FT_F32, 1e-5), (torch.float16, _lib.FT_F16, 1e-3)):
            for mode in (0, 1):
                for lpad, wpitch in ((0, W), (3, W + 6)):          # plain NHWC and the row-packed stem layout
                    n, cp = (B, 8) if mode == 0 else (2 * B, 4)
                    buf = torch.full((n, H, wpitch, cp), 3.0, device="cuda", dtype=dtype)
                    check(hip_lib.ft_flow_pack_pair(gp.data_ptr(), mean.data_ptr(), 255.0, buf.data_ptr(), B, H, W, mode, lpad, wpitch,
                                                    code, _stream()))
                    torch.cuda.synchronize()
                    got = buf.cpu().float()
                    live = got[:, :, lpad:lpad + W]
                    if mode == 0:
                        want = torch.cat((xn[:, :, 0], xn[:, :, 1]), 1).permute(0, 2, 3, 1)
                        assert (live[..., :6] - want).abs().max() <= tol and torch.all(live[..., 6:] == 0)
                    else:
                        want = torch.cat((xn[:, :, 0], xn[:, :, 1]), 0).permute(0, 2, 3, 1)
                        assert (live[..., :3] - want).abs().max() <= tol and torch.all(live[..., 3:] == 0)
                    assert torch.all(got[:, :, :lpad] == 0) and torch.all(got[:, :, lpad + W:] == 0)


# ---- networks -------------------------------------------------------------------------------------
def _build(cls, seed, dtype, **kw):
    m = cls(ARGS, **kw)
    sd = synth.fill_flow_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    m.compute_dtype = dtype
    return m, sd


def test_flownet2s_fp32_matches_reference_golden(hip_lib):
    m, sd = _build(models.FlowNet2S, SEED, torch.float32)
    B, H, W = (int(v) for v in G["synth_shape"])
    pair = synth.frame_pairs(SEED, B, H, W)
    flow = m(pair.cuda()).cpu().numpy()
    err = np.abs(flow - G["synth_flow"]).max()
    assert err <= 1e-3, f"synthetic pair: max abs err {err:.3e} px"
    # the reference's real demo pair (centre crop), demo.py:83-90 packing
    ims = torch.from_numpy(G["sample_pair_u8"][None].transpose(0, 4, 1, 2, 3).astype(np.float32))
    flow = m(ims.cuda()).cpu().numpy()
    err = np.abs(flow - G["sample_flow"]).max()
    assert err <= 1e-3, f"sample pair: max abs err {err:.3e} px"


def test_flownet2s_batchnorm_variant(hip_lib):
    m, sd = _build(models.FlowNet2S, SEED + 1, torch.float32, batchNorm=True)
    B, H, W = (int(v) for v in G["synth_shape"])
    flow = m(synth.frame_pairs(SEED, B, H, W).cuda()).cpu().numpy()
    err = np.abs(flow - G["synth_flow_bn"]).max()
    assert err <= 1e-3, f"FlowNet2S batchNorm: max abs err {err:.3e} px vs the north_star bar 1e-3"


@pytest.mark.parametrize("bn", [False, True], ids=["plain", "batchnorm"])
def test_flownet2sd_fp32_matches_reference_golden(hip_lib, bn):
    """FlowNet2SD (models.py:294-344): stride-1 stem + inter_conv decoder, vs the imported reference's output."""
    m, sd = _build(models.FlowNet2SD, SEED + 2 + int(bn), torch.float32, batchNorm=bn)
    B, H, W = (int(v) for v in G["synth_shape"])
    flow = m(synth.frame_pairs(SEED, B, H, W).cuda()).cpu().numpy()
    err = np.abs(flow - G["synth_flow_sd_bn" if bn else "synth_flow_sd"]).max()
    assert err <= 1e-3, f"FlowNet2SD bn={bn}: max abs err {err:.3e} px vs the north_star bar 1e-3"


_ORACLE_FWD = {"FlowNet2S": flow_ref.flownet2s_forward, "FlowNet2C": flow_ref.flownet2c_forward,
               "FlowNet2CS": flow_ref.flownet2cs_forward, "FlowNet2SD": flow_ref.flownet2sd_forward,
               "FlowNet2CSS": flow_ref.flownet2css_forward, "FlowNet2": flow_ref.flownet2_forward}


def test_fusion_concat_and_nearest_upsample(hip_lib, oracle_lib):
    """ft_upsample_nearest4x and ft_flow_fusion_concat (FlowNetFusion's 11-channel input, models.py:140-168)
    against Resample2d / ChannelNorm / cat of the oracle."""
    import ctypes
    from flowtrack.pytorch_amd.hip_ops import new_rowpacked_act
    B, H, W = 2, 24, 40
    x = synth.normal(5, "x6", (B, 6, H, W)) * 0.5
    fsd = synth.normal(5, "fsd", (B, 2, H, W)) * 3.0
    fs2 = synth.normal(5, "fs2", (B, 2, H, W)) * 3.0
    small = synth.normal(5, "small", (B, 2, 3, 5))
    gs, gy = _cuda(small), torch.empty((B, 2, 12, 20), dtype=torch.float32, device="cuda")
    check(hip_lib.ft_upsample_nearest4x(gs.data_ptr(), gy.data_ptr(), B, 2, 3, 5, ctypes.c_float(0.05), _stream()))
    assert torch.equal(gy.cpu(), torch.nn.functional.interpolate(small * np.float32(0.05), scale_factor=4, mode="nearest"))
    chn = lambda t: torch.from_numpy(ops_ref.channelnorm_c(t.contiguous().numpy()))
    warp = lambda f: torch.from_numpy(ops_ref.resample2d_c(x[:, 3:].contiguous().numpy(), f.contiguous().numpy()))
    want = torch.cat((x[:, :3], fsd, fs2, chn(fsd), chn(fs2), chn(x[:, :3] - warp(fsd)), chn(x[:, :3] - warp(fs2))), 1)
    for dtype, tol in ((torch.float32, 1e-5), (torch.float16, 2e-2)):
        x6 = new_rowpacked_act(B, H, W, 6, 3, dtype, "cuda")
        x6.t[:, :, x6.lpad:x6.lpad + W, :6] = x.permute(0, 2, 3, 1).to(dtype).cuda()
        y = new_rowpacked_act(B, H, W, 11, 1, dtype, "cuda")
        gsd, gs2 = _cuda(fsd), _cuda(fs2)
        check(hip_lib.ft_flow_fusion_concat(x6.t.data_ptr(), gsd.data_ptr(), gs2.data_ptr(), y.t.data_ptr(), B, H, W, x6.lpad,
                                            x6.wpitch, y.lpad, y.wpitch, _lib.dtype_code(dtype), _stream()))
        got = y.t.float().cpu()
        live = got[:, :, y.lpad:y.lpad + W]
        assert (live[..., :11].permute(0, 3, 1, 2) - want).abs().max() <= tol
        assert torch.all(live[..., 11:] == 0) and torch.all(got[:, :, :y.lpad] == 0) and torch.all(got[:, :, y.lpad + W:] == 0)


@pytest.mark.parametrize("name", ["FlowNet2C", "FlowNet2CS", "FlowNet2CSS", "FlowNet2"])
def test_flownet2c_cs_fp32_vs_oracle(hip_lib, oracle_lib, name):
    m, sd = _build(getattr(models, name), SEED + 2, torch.float32)
    pair = synth.frame_pairs(SEED + 2, 2, 128, 192)
    flow = m(pair.cuda()).cpu()
    want = _ORACLE_FWD[name](sd, pair)
    err = (flow - want).abs().max().item()
    print(name, "flow range", want.min().item(), want.max().item(), "err", err)
    assert err <= 1e-3, f"{name}: max abs err {err:.3e} px"


_C_FAMILY_GOLDEN = {"FlowNet2C": "c", "FlowNet2CS": "cs", "FlowNet2CSS": "css", "FlowNet2": "full"}


@pytest.mark.parametrize("name", list(_C_FAMILY_GOLDEN))
def test_flownet2_c_family_fp32_matches_reference_graph_golden(hip_lib, name):
    """F3 / N4: the flows the IMPORTED reference graphs produced (FlowNetC.py:71-128, models.py:108-178,180-246,346-498)
    with the restated operators injected at their FFI boundary (tests/golden/make_golden.py: inject_restated_cuda_ops).
    Pins concat orders, div_flow handling, which frame is warped and the bilinear / nearest x4 of every stacked model."""
    tag = _C_FAMILY_GOLDEN[name]
    seed = int(G[f"synth_flow_{tag}_seed"])
    m, sd = _build(getattr(models, name), seed, torch.float32)
    B, H, W = (int(v) for v in G["synth_shape"])
    flow = m(synth.frame_pairs(SEED, B, H, W).cuda()).cpu().numpy()
    want = G[f"synth_flow_{tag}"]
    err = np.abs(flow - want).max()
    print(name, "flow range", want.min(), want.max(), "err", err)
    assert err <= 1e-3, f"{name}: max abs err {err:.3e} px vs the imported reference graph"


@pytest.mark.parametrize("name", ["FlowNet2S", "FlowNet2C", "FlowNet2CS", "FlowNet2SD", "FlowNet2CSS", "FlowNet2"])
def test_flownet_fp16_vs_fp32_oracle(hip_lib, oracle_lib, name):
    """pseudo-fp16 mode (fp16 storage, fp32 accumulate; tools/flownet/demo.py --fp16): EPE vs the fp32 oracle."""
    m, sd = _build(getattr(models, name), SEED + 3, torch.float16)
    pair = synth.frame_pairs(SEED + 3, 1, 128, 192)
    flow = m(pair.cuda()).cpu()
    want = _ORACLE_FWD[name](sd, pair)
    e = flow_ref.epe(flow, want)
    mag = torch.norm(want, dim=1).mean().item()
    print(name, "fp16 EPE", e, "mean |flow|", mag)
    assert e <= 0.02 * max(mag, 1.0) + 0.05


def test_model_half_is_fp16_mode(hip_lib):
    """model.half() (demo.py:69-70) selects the fp16 path; output stays fp32 [B,2,H,W]."""
    m, sd = _build(models.FlowNet2S, SEED, torch.float32)
    m.compute_dtype = None
    m = m.half()
    pair = synth.frame_pairs(SEED, 1, 64, 64)
    out = m(pair.cuda())
    assert out.dtype == torch.float32 and tuple(out.shape) == (1, 2, 64, 64) and torch.isfinite(out).all()
