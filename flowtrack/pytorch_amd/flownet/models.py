"""FlowNet2-family optical flow on MI355X: FlowNet2S, FlowNet2C, FlowNet2CS as fused HIP launches.

Drop-in surface (reference: lib/flownet/model/models.py:180-292,346-409; trunks
lib/flownet/networks/FlowNetS.py:15-94, FlowNetC.py:13-128; builders submodules.py:7-38):
    models.FlowNet2S(args, batchNorm=False, div_flow=20)      args: .rgb_max, .fp16
    models.FlowNet2C(args, batchNorm=False, div_flow=20)
    models.FlowNet2CS(args, batchNorm=False, div_flow=20.)
    module.forward(inputs[B,3,2,H,W] in 0..rgb_max) -> flow [B,2,H,W] (fp32)
with the reference's state_dict keys so NVIDIA flownet2-pytorch checkpoints
(`ckpt['state_dict']`, tools/flownet/demo.py:52-54) load unchanged.

What differs by design: conv+bias+LeakyReLU (or folded BN) is one launch; every `torch.cat` is a
channel slice of a pre-allocated NHWC buffer that its producers write directly; FlowNetC's siamese
trunk runs once on a 2B batch; correlation reads the NHWC features and writes its LeakyReLU'd cost
volume into the conv3_1 input buffer; warp + brightness-error + concat between stacked nets is one
kernel; the x4 bilinear upsample folds in div_flow.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.nn as nn

from .. import _lib
from ..engine import HipModule
from ..hip_ops import (ActView, FlowtrackHipError, FusedConv, Program, act_stride, new_act, new_rowpacked_act,
                       record_upsample4x, round_up)
from ..params import ActMarker, BatchNormParams, ConvParams, ConvTransposeParams

LEAK = 0.1


# ---- parameter layout (submodules.py:7-38) -------------------------------------------------------
def conv(batchNorm: bool, in_planes: int, out_planes: int, kernel_size: int = 3, stride: int = 1) -> nn.Sequential:
    pad = (kernel_size - 1) // 2
    if batchNorm:
        return nn.Sequential(ConvParams(in_planes, out_planes, kernel_size, stride, pad, bias=False),
                             BatchNormParams(out_planes), ActMarker("leaky", LEAK))
    return nn.Sequential(ConvParams(in_planes, out_planes, kernel_size, stride, pad, bias=True),
                         ActMarker("leaky", LEAK))


def predict_flow(in_planes: int) -> ConvParams:
    return ConvParams(in_planes, 2, 3, 1, 1, bias=True)


def deconv(in_planes: int, out_planes: int) -> nn.Sequential:
    return nn.Sequential(ConvTransposeParams(in_planes, out_planes, bias=True), ActMarker("leaky", LEAK))


def _reference_init(module: nn.Module) -> None:
    """FlowNetS.py:47-56: xavier_uniform weights, U(0,1) biases for every conv / transposed conv."""
    for m in module.modules():
        if isinstance(m, (ConvParams, ConvTransposeParams)):
            if m.bias is not None:
                nn.init.uniform_(m.bias)
            nn.init.xavier_uniform_(m.weight)


class _Decoder(nn.Module):
    """Layers shared by FlowNetS and FlowNetC from deconv5 down (FlowNetS.py:31-45)."""

    def _make_decoder(self, upflow_bias: bool) -> None:
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(1026)
        self.predict_flow4 = predict_flow(770)
        self.predict_flow3 = predict_flow(386)
        self.predict_flow2 = predict_flow(194)
        self.upsampled_flow6_to_5 = ConvTransposeParams(2, 2, bias=upflow_bias)
        self.upsampled_flow5_to_4 = ConvTransposeParams(2, 2, bias=upflow_bias)
        self.upsampled_flow4_to_3 = ConvTransposeParams(2, 2, bias=upflow_bias)
        self.upsampled_flow3_to_2 = ConvTransposeParams(2, 2, bias=upflow_bias)


class FlowNetS(_Decoder):
    """Parameter layout of FlowNetS.py:16-45 (holders only)."""

    def __init__(self, args=None, input_channels: int = 12, batchNorm: bool = True):
        super().__init__()
        self.batchNorm = batchNorm
        self.input_channels = input_channels
        self.conv1 = conv(batchNorm, input_channels, 64, kernel_size=7, stride=2)
        self.conv2 = conv(batchNorm, 64, 128, kernel_size=5, stride=2)
        self.conv3 = conv(batchNorm, 128, 256, kernel_size=5, stride=2)
        self.conv3_1 = conv(batchNorm, 256, 256)
        self.conv4 = conv(batchNorm, 256, 512, stride=2)
        self.conv4_1 = conv(batchNorm, 512, 512)
        self.conv5 = conv(batchNorm, 512, 512, stride=2)
        self.conv5_1 = conv(batchNorm, 512, 512)
        self.conv6 = conv(batchNorm, 512, 1024, stride=2)
        self.conv6_1 = conv(batchNorm, 1024, 1024)
        self._make_decoder(upflow_bias=False)
        _reference_init(self)
        self.upsample1 = ActMarker("upsample_bilinear_x4")


class FlowNetC(_Decoder):
    """Parameter layout of FlowNetC.py:14-57 (holders only)."""

    def __init__(self, args=None, batchNorm: bool = True, div_flow: float = 20):
        super().__init__()
        self.batchNorm = batchNorm
        self.div_flow = div_flow
        self.conv1 = conv(batchNorm, 3, 64, kernel_size=7, stride=2)
        self.conv2 = conv(batchNorm, 64, 128, kernel_size=5, stride=2)
        self.conv3 = conv(batchNorm, 128, 256, kernel_size=5, stride=2)
        self.conv_redir = conv(batchNorm, 256, 32, kernel_size=1, stride=1)
        # Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2), FlowNetC.py:25-31
        self.corr = ActMarker("correlation(pad=20,k=1,d=20,s1=1,s2=2)")
        self.corr_activation = ActMarker("leaky", LEAK)
        self.conv3_1 = conv(batchNorm, 473, 256)
        self.conv4 = conv(batchNorm, 256, 512, stride=2)
        self.conv4_1 = conv(batchNorm, 512, 512)
        self.conv5 = conv(batchNorm, 512, 512, stride=2)
        self.conv5_1 = conv(batchNorm, 512, 512)
        self.conv6 = conv(batchNorm, 512, 1024, stride=2)
        self.conv6_1 = conv(batchNorm, 1024, 1024)
        self._make_decoder(upflow_bias=True)
        _reference_init(self)
        self.upsample1 = ActMarker("upsample_bilinear_x4")


def i_conv(batchNorm: bool, in_planes: int, out_planes: int, kernel_size: int = 3, stride: int = 1, bias: bool = True):
    """submodules.py:20-29: Conv2d (+BN), NO activation."""
    pad = (kernel_size - 1) // 2
    if batchNorm:
        return nn.Sequential(ConvParams(in_planes, out_planes, kernel_size, stride, pad, bias=bias), BatchNormParams(out_planes))
    return nn.Sequential(ConvParams(in_planes, out_planes, kernel_size, stride, pad, bias=bias))


class FlowNetSD(nn.Module):
    """Parameter layout of FlowNetSD.py:11-66 (holders only)."""

    def __init__(self, args=None, batchNorm: bool = True):
        super().__init__()
        self.batchNorm = batchNorm
        self.conv0 = conv(batchNorm, 6, 64)
        self.conv1 = conv(batchNorm, 64, 64, stride=2)
        self.conv1_1 = conv(batchNorm, 64, 128)
        self.conv2 = conv(batchNorm, 128, 128, stride=2)
        self.conv2_1 = conv(batchNorm, 128, 128)
        self.conv3 = conv(batchNorm, 128, 256, stride=2)
        self.conv3_1 = conv(batchNorm, 256, 256)
        self.conv4 = conv(batchNorm, 256, 512, stride=2)
        self.conv4_1 = conv(batchNorm, 512, 512)
        self.conv5 = conv(batchNorm, 512, 512, stride=2)
        self.conv5_1 = conv(batchNorm, 512, 512)
        self.conv6 = conv(batchNorm, 512, 1024, stride=2)
        self.conv6_1 = conv(batchNorm, 1024, 1024)
        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)
        self.inter_conv5 = i_conv(batchNorm, 1026, 512)
        self.inter_conv4 = i_conv(batchNorm, 770, 256)
        self.inter_conv3 = i_conv(batchNorm, 386, 128)
        self.inter_conv2 = i_conv(batchNorm, 194, 64)
        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(512)
        self.predict_flow4 = predict_flow(256)
        self.predict_flow3 = predict_flow(128)
        self.predict_flow2 = predict_flow(64)
        self.upsampled_flow6_to_5 = ConvTransposeParams(2, 2, bias=True)
        self.upsampled_flow5_to_4 = ConvTransposeParams(2, 2, bias=True)
        self.upsampled_flow4_to_3 = ConvTransposeParams(2, 2, bias=True)
        self.upsampled_flow3_to_2 = ConvTransposeParams(2, 2, bias=True)
        _reference_init(self)
        self.upsample1 = ActMarker("upsample_bilinear_x4")


class FlowNetFusion(nn.Module):
    """Parameter layout of FlowNetFusion.py:11-46 (holders only)."""

    def __init__(self, args=None, batchNorm: bool = True):
        super().__init__()
        self.batchNorm = batchNorm
        self.conv0 = conv(batchNorm, 11, 64)
        self.conv1 = conv(batchNorm, 64, 64, stride=2)
        self.conv1_1 = conv(batchNorm, 64, 128)
        self.conv2 = conv(batchNorm, 128, 128, stride=2)
        self.conv2_1 = conv(batchNorm, 128, 128)
        self.deconv1 = deconv(128, 32)
        self.deconv0 = deconv(162, 16)
        self.inter_conv1 = i_conv(batchNorm, 162, 32)
        self.inter_conv0 = i_conv(batchNorm, 82, 16)
        self.predict_flow2 = predict_flow(128)
        self.predict_flow1 = predict_flow(32)
        self.predict_flow0 = predict_flow(16)
        self.upsampled_flow2_to_1 = ConvTransposeParams(2, 2, bias=True)
        self.upsampled_flow1_to_0 = ConvTransposeParams(2, 2, bias=True)
        _reference_init(self)


CORR_MAX_DISP, CORR_STRIDE2 = 20, 2
CORR_CH = (2 * (CORR_MAX_DISP // CORR_STRIDE2) + 1) ** 2  # 441


# ---- plan builders ----------------------------------------------------------------------------
# `mk` carries dtype, device and the owning HipModule (mk["owner"].fused caches packed layers across plans)
def _mk(mk: dict) -> dict:
    return {k: v for k, v in mk.items() if k != "owner"}


def _fc(seq: nn.Sequential, label: str, mk: dict) -> FusedConv:
    c = seq[0]
    bn = seq[1].as_dict() if isinstance(seq[1], BatchNormParams) else None
    return mk["owner"].fused(label, c.weight, stride=c.stride, pad=c.padding, bias=c.bias, bn=bn, act="leaky", slope=LEAK,
                             **_mk(mk))


def _fd(seq: nn.Sequential, label: str, mk: dict) -> FusedConv:
    c = seq[0]
    return mk["owner"].fused(label, c.weight, transposed=True, stride=2, pad=1, bias=c.bias, act="leaky", slope=LEAK,
                             **_mk(mk))


def _fi(seq: nn.Sequential, label: str, mk: dict) -> FusedConv:
    """i_conv: conv (+BN) without activation."""
    c = seq[0]
    bn = seq[1].as_dict() if len(seq) > 1 and isinstance(seq[1], BatchNormParams) else None
    return mk["owner"].fused(label, c.weight, stride=c.stride, pad=c.padding, bias=c.bias, bn=bn, act=None, **_mk(mk))


def _fp(c: ConvParams, label: str, mk: dict) -> FusedConv:
    return mk["owner"].fused(label, c.weight, stride=1, pad=1, bias=c.bias, act=None, **_mk(mk))


def _fu(c: ConvTransposeParams, label: str, mk: dict) -> FusedConv:
    return mk["owner"].fused(label, c.weight, transposed=True, stride=2, pad=1, bias=c.bias, act=None, **_mk(mk))


def _record_decoder(prog: Program, p: _Decoder, conv6: ActView, cc5, cc4, cc3, cc2, prefix: str, mk: dict):
    """predict_flow6 .. predict_flow2 with concat-free writes (FlowNetS.py:69-89).
    cc5..cc2: NHWC buffers [B,h,w,act_stride(1026|770|386|194)] whose channel slice 0 already holds the skip
    feature; returns flow2 as NCHW fp32 [B,2,H/4,W/4]."""
    B, dtype, device = conv6.N, mk["dtype"], mk["device"]
    # At every level predict_flow -> upsampled_flow and the deconv are independent (both read the level's feature map,
    # they write different channel slices of the next concat buffer) and each is too small to fill the GPU: they are
    # recorded as two parallel branches (Program.fork / side / join -> parallel branches of the captured graph).
    levels = ((conv6, cc5, 1024, 512, 512, p.predict_flow6, p.upsampled_flow6_to_5, p.deconv5, "6", "6_to_5", "5", 1026),
              (None, cc4, 768, 512, 256, p.predict_flow5, p.upsampled_flow5_to_4, p.deconv4, "5", "5_to_4", "4", 770),
              (None, cc3, 384, 256, 128, p.predict_flow4, p.upsampled_flow4_to_3, p.deconv3, "4", "4_to_3", "3", 386),
              (None, cc2, 192, 128, 64, p.predict_flow3, p.upsampled_flow3_to_2, p.deconv2, "3", "3_to_2", "2", 194))
    feat = conv6
    for _, cc, foff, doff, dch, pf, upf, dec, ptag, utag, dtag, ctot in levels:
        flow = new_act(B, feat.H, feat.W, 2, dtype, device)
        prog.fork()
        with prog.side():
            _fp(pf, prefix + "predict_flow" + ptag, mk).record(prog, feat, flow)
            _fu(upf, prefix + "upsampled_flow" + utag, mk).record(prog, flow, ActView(cc, 2, foff))
        _fd(dec, prefix + "deconv" + dtag, mk).record(prog, feat, ActView(cc, dch, doff))
        prog.join()
        feat = ActView(cc, ctot, 0)
    concat2 = feat

    flow2 = torch.empty((B, 2, concat2.H, concat2.W), dtype=torch.float32, device=device)
    _fp(p.predict_flow2, prefix + "predict_flow2", mk).record(prog, concat2, flow2)
    return flow2


def _concat_buffers(B: int, H: int, W: int, dtype, device, b2: int = None):
    """cc5, cc4, cc3, cc2 tensors; cc2 may carry a larger batch (siamese FlowNetC trunk)."""
    z = lambda n, h, w, c: torch.zeros((n, h, w, act_stride(c)), dtype=dtype, device=device)
    return (z(B, H // 32, W // 32, 1026), z(B, H // 16, W // 16, 770), z(B, H // 8, W // 8, 386),
            z(b2 or B, H // 4, W // 4, 194))


def record_flownets(prog: Program, p: FlowNetS, x: ActView, prefix: str, mk: dict, mean_fold=None) -> torch.Tensor:
    """FlowNetS.forward (FlowNetS.py:60-94) on an NHWC input view; returns flow2 NCHW fp32.
    mean_fold (from _FlowBase._record_pack_fold): `x` is the UN-centred, physically padded frame pair [B, H + 6, W + 6] and
    conv1 runs as a pad-0 conv on it with the per-sample shift that removes the rgb mean."""
    B, dtype, device = x.N, mk["dtype"], mk["device"]
    if mean_fold is not None:
        H, W = x.H - 6, x.W - 6
    else:
        H, W = x.H, x.W
    cc5, cc4, cc3, cc2 = _concat_buffers(B, H, W, dtype, device)
    c1 = new_act(B, H // 2, W // 2, 64, dtype, device)
    if mean_fold is not None:
        c = p.conv1[0]
        bn = p.conv1[1].as_dict() if isinstance(p.conv1[1], BatchNormParams) else None
        layer = mk["owner"].fused(prefix + "conv1.meanfold", c.weight, stride=c.stride, pad=0, bias=c.bias, bn=bn, act="leaky",
                                  slope=LEAK, **_mk(mk))
        layer.record(prog, x, c1, shift_n=mean_fold(c.weight))
    else:
        _fc(p.conv1, prefix + "conv1", mk).record(prog, x, c1)
    _fc(p.conv2, prefix + "conv2", mk).record(prog, c1, ActView(cc2, 128, 0))
    c3 = new_act(B, H // 8, W // 8, 256, dtype, device)
    _fc(p.conv3, prefix + "conv3", mk).record(prog, ActView(cc2, 128, 0), c3)
    _fc(p.conv3_1, prefix + "conv3_1", mk).record(prog, c3, ActView(cc3, 256, 0))
    c4 = new_act(B, H // 16, W // 16, 512, dtype, device)
    _fc(p.conv4, prefix + "conv4", mk).record(prog, ActView(cc3, 256, 0), c4)
    _fc(p.conv4_1, prefix + "conv4_1", mk).record(prog, c4, ActView(cc4, 512, 0))
    c5 = new_act(B, H // 32, W // 32, 512, dtype, device)
    _fc(p.conv5, prefix + "conv5", mk).record(prog, ActView(cc4, 512, 0), c5)
    _fc(p.conv5_1, prefix + "conv5_1", mk).record(prog, c5, ActView(cc5, 512, 0))
    c6 = new_act(B, H // 64, W // 64, 1024, dtype, device)
    _fc(p.conv6, prefix + "conv6", mk).record(prog, ActView(cc5, 512, 0), c6)
    c61 = new_act(B, H // 64, W // 64, 1024, dtype, device)
    _fc(p.conv6_1, prefix + "conv6_1", mk).record(prog, c6, c61)
    return _record_decoder(prog, p, c61, cc5, cc4, cc3, cc2, prefix, mk)


def record_flownetc(prog: Program, p: FlowNetC, x2b: ActView, prefix: str, mk: dict) -> torch.Tensor:
    """FlowNetC.forward (FlowNetC.py:71-128). x2b: row-packed [2B,H,W+6,4] view with 3 channels, images
    0..B-1 = frame 0 and B..2B-1 = frame 1 (the siamese conv1-3 run once on the 2B batch)."""
    B2, H, W, dtype, device = x2b.N, x2b.H, x2b.W, mk["dtype"], mk["device"]
    B = B2 // 2
    cc5, cc4, cc3, cc2 = _concat_buffers(B, H, W, dtype, device, b2=B2)
    c1 = new_act(B2, H // 2, W // 2, 64, dtype, device)
    _fc(p.conv1, prefix + "conv1", mk).record(prog, x2b, c1)
    _fc(p.conv2, prefix + "conv2", mk).record(prog, c1, ActView(cc2, 128, 0))       # out_conv2a = images 0..B-1
    c3 = new_act(B2, H // 8, W // 8, 256, dtype, device)
    _fc(p.conv3, prefix + "conv3", mk).record(prog, ActView(cc2, 128, 0), c3)
    c3a, c3b = c3.batch_slice(0, B), c3.batch_slice(B, B2)
    # in_conv3_1 = cat(conv_redir(32), leaky(corr)(441)) -> 473 channels (FlowNetC.py:86-92)
    cin31 = torch.zeros((B, H // 8, W // 8, 480), dtype=dtype, device=device)
    _fc(p.conv_redir, prefix + "conv_redir", mk).record(prog, c3a, ActView(cin31, 32, 0))
    prog.add("ft_correlation_nhwc_fwd", c3a.t.data_ptr(), c3b.t.data_ptr(), cin31.data_ptr(), B, 256, H // 8, W // 8,
             CORR_MAX_DISP, CORR_STRIDE2, c3.cstride, 480, 32, _lib.FT_ACT_LEAKY, ctypes.c_float(LEAK),
             _lib.dtype_code(dtype), keep=(c3.t, cin31))
    _fc(p.conv3_1, prefix + "conv3_1", mk).record(prog, ActView(cin31, 473, 0), ActView(cc3, 256, 0))
    c4 = new_act(B, H // 16, W // 16, 512, dtype, device)
    _fc(p.conv4, prefix + "conv4", mk).record(prog, ActView(cc3, 256, 0), c4)
    _fc(p.conv4_1, prefix + "conv4_1", mk).record(prog, c4, ActView(cc4, 512, 0))
    c5 = new_act(B, H // 32, W // 32, 512, dtype, device)
    _fc(p.conv5, prefix + "conv5", mk).record(prog, ActView(cc4, 512, 0), c5)
    _fc(p.conv5_1, prefix + "conv5_1", mk).record(prog, c5, ActView(cc5, 512, 0))
    c6 = new_act(B, H // 64, W // 64, 1024, dtype, device)
    _fc(p.conv6, prefix + "conv6", mk).record(prog, ActView(cc5, 512, 0), c6)
    c61 = new_act(B, H // 64, W // 64, 1024, dtype, device)
    _fc(p.conv6_1, prefix + "conv6_1", mk).record(prog, c6, c61)
    return _record_decoder(prog, p, c61, cc5, cc4, cc3, cc2[:B], prefix, mk)


def record_flownetsd(prog: Program, p: FlowNetSD, x: ActView, prefix: str, mk: dict) -> torch.Tensor:
    """FlowNetSD.forward (FlowNetSD.py:68-106): stride-1 stem, inter_conv before every flow prediction."""
    B, H, W, dtype, device = x.N, x.H, x.W, mk["dtype"], mk["device"]
    cc5, cc4, cc3, cc2 = _concat_buffers(B, H, W, dtype, device)
    act = lambda h, w, c: new_act(B, h, w, c, dtype, device)
    c0 = act(H, W, 64)
    _fc(p.conv0, prefix + "conv0", mk).record(prog, x, c0)
    c1 = act(H // 2, W // 2, 64)
    _fc(p.conv1, prefix + "conv1", mk).record(prog, c0, c1)
    c11 = act(H // 2, W // 2, 128)
    _fc(p.conv1_1, prefix + "conv1_1", mk).record(prog, c1, c11)
    c2 = act(H // 4, W // 4, 128)
    _fc(p.conv2, prefix + "conv2", mk).record(prog, c11, c2)
    _fc(p.conv2_1, prefix + "conv2_1", mk).record(prog, c2, ActView(cc2, 128, 0))
    c3 = act(H // 8, W // 8, 256)
    _fc(p.conv3, prefix + "conv3", mk).record(prog, ActView(cc2, 128, 0), c3)
    _fc(p.conv3_1, prefix + "conv3_1", mk).record(prog, c3, ActView(cc3, 256, 0))
    c4 = act(H // 16, W // 16, 512)
    _fc(p.conv4, prefix + "conv4", mk).record(prog, ActView(cc3, 256, 0), c4)
    _fc(p.conv4_1, prefix + "conv4_1", mk).record(prog, c4, ActView(cc4, 512, 0))
    c5 = act(H // 32, W // 32, 512)
    _fc(p.conv5, prefix + "conv5", mk).record(prog, ActView(cc4, 512, 0), c5)
    _fc(p.conv5_1, prefix + "conv5_1", mk).record(prog, c5, ActView(cc5, 512, 0))
    c6 = act(H // 64, W // 64, 1024)
    _fc(p.conv6, prefix + "conv6", mk).record(prog, ActView(cc5, 512, 0), c6)
    c61 = act(H // 64, W // 64, 1024)
    _fc(p.conv6_1, prefix + "conv6_1", mk).record(prog, c6, c61)

    flow = act(c61.H, c61.W, 2)
    _fp(p.predict_flow6, prefix + "predict_flow6", mk).record(prog, c61, flow)
    prev = c61
    stages = ((cc5, 1026, 512, 1024, p.upsampled_flow6_to_5, p.deconv5, p.inter_conv5, p.predict_flow5, "5"),
              (cc4, 770, 512, 768, p.upsampled_flow5_to_4, p.deconv4, p.inter_conv4, p.predict_flow4, "4"),
              (cc3, 386, 256, 384, p.upsampled_flow4_to_3, p.deconv3, p.inter_conv3, p.predict_flow3, "3"),
              (cc2, 194, 128, 192, p.upsampled_flow3_to_2, p.deconv2, p.inter_conv2, p.predict_flow2, "2"))
    for cc, ctot, doff, foff, upf, dec, inter, pred, tag in stages:
        up_name = {"5": "upsampled_flow6_to_5", "4": "upsampled_flow5_to_4", "3": "upsampled_flow4_to_3", "2": "upsampled_flow3_to_2"}[tag]
        _fu(upf, prefix + up_name, mk).record(prog, flow, ActView(cc, 2, foff))
        _fd(dec, prefix + "deconv" + tag, mk).record(prog, prev, ActView(cc, dec[0].cout, doff))
        concat = ActView(cc, ctot, 0)
        inter_out = act(concat.H, concat.W, inter[0].cout)
        _fi(inter, prefix + "inter_conv" + tag, mk).record(prog, concat, inter_out)
        if tag == "2":
            flow2 = torch.empty((B, 2, concat.H, concat.W), dtype=torch.float32, device=device)
            _fp(pred, prefix + "predict_flow2", mk).record(prog, inter_out, flow2)
            return flow2
        flow = act(concat.H, concat.W, 2)
        _fp(pred, prefix + "predict_flow" + tag, mk).record(prog, inter_out, flow)
        prev = concat


def record_flownetfusion(prog: Program, p: FlowNetFusion, x: ActView, prefix: str, mk: dict) -> torch.Tensor:
    """FlowNetFusion.forward (FlowNetFusion.py:48-66) on the 11-channel full-resolution input; returns flow0
    NCHW fp32 [B,2,H,W]."""
    B, H, W, dtype, device = x.N, x.H, x.W, mk["dtype"], mk["device"]
    z = lambda h, w, c: torch.zeros((B, h, w, act_stride(c)), dtype=dtype, device=device)
    cat1 = z(H // 2, W // 2, 162)      # (conv1_1 128 | deconv1 32 | flow2_up 2)
    cat0 = z(H, W, 82)                 # (conv0 64 | deconv0 16 | flow1_up 2)
    _fc(p.conv0, prefix + "conv0", mk).record(prog, x, ActView(cat0, 64, 0))
    c1 = new_act(B, H // 2, W // 2, 64, dtype, device)
    _fc(p.conv1, prefix + "conv1", mk).record(prog, ActView(cat0, 64, 0), c1)
    _fc(p.conv1_1, prefix + "conv1_1", mk).record(prog, c1, ActView(cat1, 128, 0))
    c2 = new_act(B, H // 4, W // 4, 128, dtype, device)
    _fc(p.conv2, prefix + "conv2", mk).record(prog, ActView(cat1, 128, 0), c2)
    c21 = new_act(B, H // 4, W // 4, 128, dtype, device)
    _fc(p.conv2_1, prefix + "conv2_1", mk).record(prog, c2, c21)
    flow2 = new_act(B, H // 4, W // 4, 2, dtype, device)
    _fp(p.predict_flow2, prefix + "predict_flow2", mk).record(prog, c21, flow2)
    _fu(p.upsampled_flow2_to_1, prefix + "upsampled_flow2_to_1", mk).record(prog, flow2, ActView(cat1, 2, 160))
    _fd(p.deconv1, prefix + "deconv1", mk).record(prog, c21, ActView(cat1, 32, 128))
    concat1 = ActView(cat1, 162, 0)
    i1 = new_act(B, H // 2, W // 2, 32, dtype, device)
    _fi(p.inter_conv1, prefix + "inter_conv1", mk).record(prog, concat1, i1)
    flow1 = new_act(B, H // 2, W // 2, 2, dtype, device)
    _fp(p.predict_flow1, prefix + "predict_flow1", mk).record(prog, i1, flow1)
    _fu(p.upsampled_flow1_to_0, prefix + "upsampled_flow1_to_0", mk).record(prog, flow1, ActView(cat0, 2, 80))
    _fd(p.deconv0, prefix + "deconv0", mk).record(prog, concat1, ActView(cat0, 16, 64))
    i0 = new_act(B, H, W, 16, dtype, device)
    _fi(p.inter_conv0, prefix + "inter_conv0", mk).record(prog, ActView(cat0, 82, 0), i0)
    flow0 = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
    _fp(p.predict_flow0, prefix + "predict_flow0", mk).record(prog, i0, flow0)
    return flow0


#: rgb mean + normalise + pack as one launch (ft_flow_mean_pack_pair) instead of two.  OFF by default: measured at 16 x 512 x 384 the
#: launch takes 46 us against 23 + 30 us, but the graph replay of FlowNet2S does not move (0.923-0.925 vs 0.923-0.928 ms, same box:
#: the sample's workgroups wait for each other between their read and their write phase), and a launch whose workgroups wait for
#: each other is not worth carrying for nothing.  FT_FUSE_MEAN_PACK=1 records it (tests/test_flow_gpu.py covers it either way).
FUSE_MEAN_PACK = os.environ.get("FT_FUSE_MEAN_PACK", "0") == "1"
#: FlowNet2S: the rgb mean folded into conv1 (one pass over the frames instead of mean + pack: _FlowBase._record_pack_fold).
#: FT_MEAN_FOLD=0 keeps the two launches.
MEAN_FOLD = os.environ.get("FT_MEAN_FOLD", "1") != "0"


class _FlowPlan:
    def __init__(self, prog, x_static, out):
        self.prog, self.x_static, self.out = prog, x_static, out
        self.runs = 0


class _FlowBase(HipModule):
    """Shared forward / plan cache of the FlowNet2* wrappers."""
    rgb_max: float = 255.0
    div_flow: float = 20.0

    def _record_normalise(self, prog: Program, x_static: torch.Tensor, modes, dtype, device, pad: int = 3):
        """rgb_mean + (x-mean)/rgb_max (models.py:255-257); returns one row-packed NHWC view per mode
        (mode 0: [B,H,W+2*pad,8] with 6 channels; mode 1: [2B,H,W+2*pad,4] with 3 channels; `pad` = padding of
        the conv that reads it: 3 for the 7x7/s2 stems, 1 for FlowNetSD's 3x3/s1 conv0).  A mode may also be a
        (mode, pad) pair."""
        B, _, _, H, W = x_static.shape
        mean = torch.empty((B * 3,), dtype=torch.float32, device=device)
        modes = [m if isinstance(m, tuple) else (m, pad) for m in modes]
        outs = [None] * len(modes)
        words = int(_lib.load().ft_flow_mean_pack_pair_state_words(B, H, W)) if FUSE_MEAN_PACK else 0
        if words:
            # mean + normalise + pack as ONE launch that reads the frame pair once (ft_flow_mean_pack_pair: the workgroups of a
            # sample exchange their partial sums inside the launch); it writes the first 6-channel view and the first siamese
            # view asked for, any further view (another padding) is packed from the mean it leaves behind
            i6 = next((i for i, (mode, _) in enumerate(modes) if mode == 0), None)
            i3 = next((i for i, (mode, _) in enumerate(modes) if mode == 1), None)
            v6 = new_rowpacked_act(B, H, W, 6, modes[i6][1], dtype, device) if i6 is not None else None
            v3 = new_rowpacked_act(2 * B, H, W, 3, modes[i3][1], dtype, device) if i3 is not None else None
            state = torch.zeros((words,), dtype=torch.int64, device=device)         # zeroed once; private to this plan
            prog.add("ft_flow_mean_pack_pair", x_static.data_ptr(), ctypes.c_float(self.rgb_max),
                     v6.t.data_ptr() if v6 is not None else None, v6.lpad if v6 is not None else 0, v6.wpitch if v6 is not None else 0,
                     v3.t.data_ptr() if v3 is not None else None, v3.lpad if v3 is not None else 0, v3.wpitch if v3 is not None else 0,
                     B, H, W, _lib.dtype_code(dtype), state.data_ptr(), mean.data_ptr(),
                     keep=(x_static, state, mean, v6.t if v6 is not None else None, v3.t if v3 is not None else None))
            if i6 is not None:
                outs[i6] = v6
            if i3 is not None:
                outs[i3] = v3
        else:
            partial = torch.empty((B * 3 * _lib.FT_RGB_MEAN_SPLITS,), dtype=torch.float32, device=device)
            prog.add("ft_flow_rgb_mean", x_static.data_ptr(), B, H, W, partial.data_ptr(), mean.data_ptr(),
                     keep=(x_static, partial, mean))
        for i, (mode, vpad) in enumerate(modes):
            if outs[i] is not None:
                continue
            view = new_rowpacked_act(B if mode == 0 else 2 * B, H, W, 6 if mode == 0 else 3, vpad, dtype, device)
            prog.add("ft_flow_pack_pair", x_static.data_ptr(), mean.data_ptr(), ctypes.c_float(self.rgb_max),
                     view.t.data_ptr(), B, H, W, mode, view.lpad, view.wpitch, _lib.dtype_code(dtype), keep=(view.t,))
            outs[i] = view
        return outs

    def _record_pack_fold(self, prog: Program, x_static: torch.Tensor, dtype, device, pad: int = 3):
        """The rgb mean folded into conv1 (include/flowtrack_hip.h, ft_flow_pack_pair_sums / ft_flow_mean_fold): ONE pass over the
        frames writes x / rgb_max into a row- and column-padded view [B, H + 2 pad, W + 2 pad, 8] and the colour sums; returns
        (view, hook) where hook(conv1_weight) is FusedConv.record's `shift_n` callable: it records the launch that finishes the
        mean, fills the view's padding with it and writes the per-sample shift, right in front of conv1.  None when the shape
        is not covered (fp16, W % 4 == 0 only)."""
        B, _, _, H, W = x_static.shape
        if dtype != torch.float16 or W % 4:
            return None
        lib = _lib.load()
        Hp, Wp = H + 2 * pad, W + 2 * pad
        wpitch = round_up(Wp, 2)
        view = ActView(torch.zeros((B, Hp, wpitch, 8), dtype=dtype, device=device), 6, 0, 0, Wp)
        nchunk = int(lib.ft_flow_pack_pair_sums_chunks(H))
        partial = torch.empty((B * 3 * nchunk,), dtype=torch.float32, device=device)
        mean = torch.empty((B * 3,), dtype=torch.float32, device=device)
        prog.add("ft_flow_pack_pair_sums", x_static.data_ptr(), ctypes.c_float(self.rgb_max), view.t.data_ptr(), B, H, W, pad,
                 wpitch, _lib.dtype_code(dtype), partial.data_ptr(), keep=(x_static, view.t, partial))

        def hook_for(weight: torch.Tensor):
            # the kernel window summed per input channel, over the fp16 values the matrix pipe multiplies
            wsum = torch.zeros((weight.shape[0], 8), dtype=torch.float32)
            wsum[:, :weight.shape[1]] = weight.detach().to(torch.float32).cpu().to(dtype).to(torch.float64).sum(dim=(2, 3)).float()
            wsum = wsum.to(device)
            shift_n = torch.empty((B, weight.shape[0]), dtype=torch.float32, device=device)

            def hook(scale, shift):
                prog.add("ft_flow_mean_fold", partial.data_ptr(), ctypes.c_float(self.rgb_max), view.t.data_ptr(), B, H, W, pad, wpitch,
                         _lib.dtype_code(dtype), wsum.data_ptr(), scale.data_ptr() if scale is not None else None,
                         shift.data_ptr() if shift is not None else None, weight.shape[0], shift_n.data_ptr(), mean.data_ptr(),
                         keep=(partial, view.t, wsum, scale, shift, shift_n, mean))
                return shift_n
            return hook
        return view, hook_for

    def _build_plan(self, B, H, W, device, dtype) -> _FlowPlan:  # pragma: no cover - abstract
        raise NotImplementedError

    def plan_for(self, B: int, H: int, W: int, replica: int = 0) -> _FlowPlan:
        """`replica` > 0: an independent copy of the plan (own input / activation buffers and graph, same packed weights) for
        callers that keep several batches of one shape resident (bench.py's rotation; as DeconvResnet.plan_for)."""
        device, dtype = self._resolve()
        if H % 64 or W % 64:
            raise FlowtrackHipError(f"frame size {H}x{W}: FlowNet needs multiples of 64")
        key = (B, H, W, device, dtype) + ((replica,) if replica else ())
        plan = self._plans.get(key)
        if plan is None:
            with torch.no_grad():
                plan = self._build_plan(B, H, W, device, dtype)
            self._plans[key] = plan
        return plan

    def static_input(self, B: int, H: int, W: int) -> torch.Tensor:
        """The plan's own input buffer [B,3,2,H,W] fp32 (the fixed address its graph reads): write the batch here and
        pass this tensor to forward() to skip the staging copy (75 MB at 16 x 512x384)."""
        return self.plan_for(B, H, W).x_static

    @torch.no_grad()
    def forward(self, inputs: torch.Tensor, copy_output: bool = True) -> torch.Tensor:
        """inputs [B,3,2,H,W] (RGB, 0..rgb_max) -> flow [B,2,H,W] fp32 in pixels."""
        self._check_eval()
        if inputs.dim() != 5 or inputs.shape[1] != 3 or inputs.shape[2] != 2:
            raise FlowtrackHipError(f"expected [B,3,2,H,W], got {tuple(inputs.shape)}")
        B, _, _, H, W = inputs.shape
        plan = self.plan_for(B, H, W)
        if inputs.device != plan.x_static.device:
            raise FlowtrackHipError("input and model are on different devices")
        if inputs.data_ptr() != plan.x_static.data_ptr():   # zero-copy when the caller filled static_input() in place
            plan.x_static.copy_(inputs)
        self._run_plan(plan.prog, first=plan.runs == 0)
        plan.runs += 1
        return plan.out.clone() if copy_output else plan.out

    @torch.no_grad()
    def replay(self, plan: _FlowPlan) -> _FlowPlan:
        """Run a plan whose static input the caller filled in place (plan_for(...).x_static); the flow is plan.out."""
        self._check_eval()
        self._run_plan(plan.prog, first=plan.runs == 0)
        plan.runs += 1
        return plan


class FlowNet2S(FlowNetS, _FlowBase):
    def __init__(self, args, batchNorm: bool = False, div_flow: float = 20):
        _FlowBase.__init__(self)
        FlowNetS.__init__(self, args, input_channels=6, batchNorm=batchNorm)
        self.rgb_max = float(args.rgb_max)
        self.div_flow = float(div_flow)

    def _build_plan(self, B, H, W, device, dtype) -> _FlowPlan:
        prog = Program(self._side_stream(device))
        mk = dict(dtype=dtype, device=device, owner=self)
        x_static = torch.empty((B, 3, 2, H, W), dtype=torch.float32, device=device)
        fold = None
        if MEAN_FOLD and self._fold_supported(B, H, W, dtype):
            fold = self._record_pack_fold(prog, x_static, dtype, device)
        if fold is not None:
            flow2 = record_flownets(prog, self, fold[0], "", mk, mean_fold=fold[1])
        else:
            (x6,) = self._record_normalise(prog, x_static, (0,), dtype, device)
            flow2 = record_flownets(prog, self, x6, "", mk)
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample4x(prog, flow2, out, self.div_flow)  # upsample1(flow2 * div_flow), models.py:292
        return _FlowPlan(prog, x_static, out)

    def _fold_supported(self, B: int, H: int, W: int, dtype, pad: int = 3) -> bool:
        """Whether conv1 on the padded view runs on the one kernel that takes a per-sample shift (fp16, large enough grids).
        NOTE: the fold therefore depends on the BATCH — one 384 x 512 pair (384 stem tiles) takes the two-launch path
        (mean, then pack of the centred input), two or more pairs take the fold, which rounds the input to fp16 BEFORE the mean is
        removed and enters mean / rgb_max as its fp16 value.  The flows of one pair at batch 1 and at batch >= 2 so differ at
        fp16-rounding level (both within the fp16 mode's EPE bound, tests/test_flow_gpu.py); FT_MEAN_FOLD=0 pins the two-launch
        path for every batch size."""
        if dtype != torch.float16 or W % 4:
            return False
        d = _lib.ConvDesc()
        d.dtype = _lib.dtype_code(dtype)
        d.N, d.Hi, d.Wi, d.Cin, d.x_cstride = B, H + 2 * pad, W + 2 * pad, 6, 8
        d.x_wpitch = round_up(W + 2 * pad, 2)
        d.Cout, d.kh, d.kw, d.stride, d.pad = 64, 7, 7, 2, 0
        d.Ho, d.Wo = H // 2, W // 2
        d.y_cstride, d.out_layout = act_stride(64), _lib.FT_LAYOUT_NHWC
        d.act, d.slope = _lib.FT_ACT_LEAKY, LEAK
        d.shift_nstride = 64
        return _lib.load().ft_conv_shift_nstride_supported(ctypes.byref(d)) == 0


class FlowNet2C(FlowNetC, _FlowBase):
    def __init__(self, args, batchNorm: bool = False, div_flow: float = 20):
        _FlowBase.__init__(self)
        FlowNetC.__init__(self, args, batchNorm=batchNorm, div_flow=20)  # models.py:182 pins 20
        self.rgb_max = float(args.rgb_max)
        self.div_flow = 20.0

    def _build_plan(self, B, H, W, device, dtype) -> _FlowPlan:
        prog = Program(self._side_stream(device))
        mk = dict(dtype=dtype, device=device, owner=self)
        x_static = torch.empty((B, 3, 2, H, W), dtype=torch.float32, device=device)
        (x2b,) = self._record_normalise(prog, x_static, (1,), dtype, device)
        flow2 = record_flownetc(prog, self, x2b, "", mk)
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample4x(prog, flow2, out, self.div_flow)
        return _FlowPlan(prog, x_static, out)


class FlowNet2CS(_FlowBase):
    def __init__(self, args, batchNorm: bool = False, div_flow: float = 20.):
        super().__init__()
        self.batchNorm = batchNorm
        self.div_flow = float(div_flow)
        self.rgb_max = float(args.rgb_max)
        self.args = args
        self.channelnorm = ActMarker("channelnorm")
        self.flownetc = FlowNetC(args, batchNorm=batchNorm)
        self.upsample1 = ActMarker("upsample_bilinear_x4")
        self.resample1 = ActMarker("resample2d")
        self.flownets_1 = FlowNetS(args, batchNorm=batchNorm)
        self.upsample2 = ActMarker("upsample_bilinear_x4")

    def _build_plan(self, B, H, W, device, dtype) -> _FlowPlan:
        prog = Program(self._side_stream(device))
        mk = dict(dtype=dtype, device=device, owner=self)
        x_static = torch.empty((B, 3, 2, H, W), dtype=torch.float32, device=device)
        x6, x2b = self._record_normalise(prog, x_static, (0, 1), dtype, device)
        flow2c = record_flownetc(prog, self.flownetc, x2b, "flownetc.", mk)
        flowc = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample4x(prog, flow2c, flowc, self.div_flow)           # models.py:393-394
        concat1 = new_rowpacked_act(B, H, W, 12, 3, dtype, device)
        prog.add("ft_flow_warp_concat", x6.t.data_ptr(), flowc.data_ptr(), ctypes.c_float(self.div_flow),
                 concat1.t.data_ptr(), B, H, W, x6.lpad, x6.wpitch, concat1.lpad, concat1.wpitch,
                 _lib.dtype_code(dtype), keep=(x6.t, flowc, concat1.t))                           # models.py:396-403
        flow2s = record_flownets(prog, self.flownets_1, concat1, "flownets_1.", mk)
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample4x(prog, flow2s, out, self.div_flow)             # models.py:406-407
        return _FlowPlan(prog, x_static, out)


def record_upsample_nearest4x(prog: Program, x: torch.Tensor, y: torch.Tensor, mul: float) -> None:
    N, C, h, w = x.shape
    prog.add("ft_upsample_nearest4x", x.data_ptr(), y.data_ptr(), N, C, h, w, ctypes.c_float(mul), keep=(x, y))


def _record_warp_stage(prog: Program, x6: ActView, flow: torch.Tensor, div_flow: float, dtype, device) -> ActView:
    """(img0, img1, warp(img1, flow), flow/div_flow, |img0 - warp|) -> row-packed 12-channel input of the next FlowNetS."""
    B, H, W = x6.N, x6.H, x6.W
    concat = new_rowpacked_act(B, H, W, 12, 3, dtype, device)
    prog.add("ft_flow_warp_concat", x6.t.data_ptr(), flow.data_ptr(), ctypes.c_float(div_flow), concat.t.data_ptr(),
             B, H, W, x6.lpad, x6.wpitch, concat.lpad, concat.wpitch, _lib.dtype_code(dtype), keep=(x6.t, flow, concat.t))
    return concat


class FlowNet2SD(FlowNetSD, _FlowBase):
    """models.py:294-344."""

    def __init__(self, args, batchNorm: bool = False, div_flow: float = 20):
        _FlowBase.__init__(self)
        FlowNetSD.__init__(self, args, batchNorm=batchNorm)
        self.rgb_max = float(args.rgb_max)
        self.div_flow = float(div_flow)

    def _build_plan(self, B, H, W, device, dtype) -> _FlowPlan:
        prog = Program(self._side_stream(device))
        mk = dict(dtype=dtype, device=device, owner=self)
        x_static = torch.empty((B, 3, 2, H, W), dtype=torch.float32, device=device)
        (x6,) = self._record_normalise(prog, x_static, (0,), dtype, device, pad=1)     # conv0 is 3x3 / s1 / p1
        flow2 = record_flownetsd(prog, self, x6, "", mk)
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample4x(prog, flow2, out, self.div_flow)
        return _FlowPlan(prog, x_static, out)


class FlowNet2CSS(_FlowBase):
    """models.py:411-498: FlowNetC -> warp -> FlowNetS -> warp -> FlowNetS, last upsample NEAREST."""

    def __init__(self, args, batchNorm: bool = False, div_flow: float = 20.):
        super().__init__()
        self.batchNorm = batchNorm
        self.div_flow = float(div_flow)
        self.rgb_max = float(args.rgb_max)
        self.args = args
        self.channelnorm = ActMarker("channelnorm")
        self.flownetc = FlowNetC(args, batchNorm=batchNorm)
        self.upsample1 = ActMarker("upsample_bilinear_x4")
        self.resample1 = ActMarker("resample2d")
        self.flownets_1 = FlowNetS(args, batchNorm=batchNorm)
        self.upsample2 = ActMarker("upsample_bilinear_x4")
        self.resample2 = ActMarker("resample2d")
        self.flownets_2 = FlowNetS(args, batchNorm=batchNorm)
        self.upsample3 = ActMarker("upsample_nearest_x4")

    def _record_css(self, prog, x_static, B, H, W, dtype, device, mk, extra_modes=()):
        """Shared with FlowNet2: returns (x6, flownets2_flow2 NCHW fp32 at 1/4 resolution, extra normalised views)."""
        x6, x2b, *extra = self._record_normalise(prog, x_static, (0, 1) + tuple(extra_modes), dtype, device)
        flow2c = record_flownetc(prog, self.flownetc, x2b, "flownetc.", mk)
        flowc = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample4x(prog, flow2c, flowc, self.div_flow)
        concat1 = _record_warp_stage(prog, x6, flowc, self.div_flow, dtype, device)
        flow2s1 = record_flownets(prog, self.flownets_1, concat1, "flownets_1.", mk)
        flows1 = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample4x(prog, flow2s1, flows1, self.div_flow)
        concat2 = _record_warp_stage(prog, x6, flows1, self.div_flow, dtype, device)
        flow2s2 = record_flownets(prog, self.flownets_2, concat2, "flownets_2.", mk)
        return x6, flow2s2, extra

    def _build_plan(self, B, H, W, device, dtype) -> _FlowPlan:
        prog = Program(self._side_stream(device))
        mk = dict(dtype=dtype, device=device, owner=self)
        x_static = torch.empty((B, 3, 2, H, W), dtype=torch.float32, device=device)
        _, flow2s2, _ = self._record_css(prog, x_static, B, H, W, dtype, device, mk)
        out = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample_nearest4x(prog, flow2s2, out, self.div_flow)                   # models.py:495-496
        return _FlowPlan(prog, x_static, out)


class FlowNet2(FlowNet2CSS):
    """models.py:19-178: the CSS stack + FlowNetSD on the raw pair + FlowNetFusion.  Inference only: the
    reference's gradient hooks (`register_hook(save_grad(...))`, :151-176) have no counterpart."""

    def __init__(self, args, batchNorm: bool = False, div_flow: float = 20.):
        super().__init__(args, batchNorm=batchNorm, div_flow=div_flow)
        del self.upsample3
        self.flownets_d = FlowNetSD(args, batchNorm=batchNorm)
        self.upsample3 = ActMarker("upsample_nearest_x4")
        self.upsample4 = ActMarker("upsample_nearest_x4")
        self.resample3 = ActMarker("resample2d")
        self.resample4 = ActMarker("resample2d")
        self.flownetfusion = FlowNetFusion(args, batchNorm=batchNorm)

    def _build_plan(self, B, H, W, device, dtype) -> _FlowPlan:
        prog = Program(self._side_stream(device))
        mk = dict(dtype=dtype, device=device, owner=self)
        x_static = torch.empty((B, 3, 2, H, W), dtype=torch.float32, device=device)
        # FlowNetSD reads the same normalised pair through a 3x3/s1/p1 stem: its own row-packed copy (pad 1)
        x6, flow2s2, (x6sd,) = self._record_css(prog, x_static, B, H, W, dtype, device, mk, extra_modes=((0, 1),))
        flows2 = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample_nearest4x(prog, flow2s2, flows2, self.div_flow)                # upsample4(flow2 * div_flow), :131
        flow2sd = record_flownetsd(prog, self.flownets_d, x6sd, "flownets_d.", mk)
        flowsd = torch.empty((B, 2, H, W), dtype=torch.float32, device=device)
        record_upsample_nearest4x(prog, flow2sd, flowsd, 1.0 / self.div_flow)         # upsample3(flow2 / div_flow), :145
        cat3 = new_rowpacked_act(B, H, W, 11, 1, dtype, device)                        # fusion conv0 is 3x3/s1/p1
        prog.add("ft_flow_fusion_concat", x6.t.data_ptr(), flowsd.data_ptr(), flows2.data_ptr(), cat3.t.data_ptr(), B, H, W,
                 x6.lpad, x6.wpitch, cat3.lpad, cat3.wpitch, _lib.dtype_code(dtype), keep=(x6.t, flowsd, flows2, cat3.t))   # :134-168
        out = record_flownetfusion(prog, self.flownetfusion, cat3, "flownetfusion.", mk)
        return _FlowPlan(prog, x_static, out)
