"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional torch-CPU fp32 restatement of the FlowNet2 family, driven by reference-keyed state_dicts:
  flownet2s_forward   <- FlowNet2S.forward   lib/flownet/model/models.py:254-292
  flownet2c_forward   <- FlowNet2C.forward   models.py:185-246  (trunk FlowNetC.py:71-128)
  flownet2cs_forward  <- FlowNet2CS.forward  models.py:383-409
  flownets_trunk      <- FlowNetS.forward    lib/flownet/networks/FlowNetS.py:60-94
  flownet2sd_forward  <- FlowNet2SD.forward  models.py:330-344  (trunk FlowNetSD.py:68-106)
  flownet2css_forward <- FlowNet2CSS.forward models.py:455-498
  flownet2_forward    <- FlowNet2.forward    models.py:108-178  (fusion FlowNetFusion.py:48-66)
Stock layers are torch.nn.functional on CPU (what the reference itself executes); the three
CUDA-only operators come from oracle/ops_ref.py (C restatement of the .cu kernels).
FlowNet2S and FlowNet2SD are pinned against the imported reference by tests/golden/make_golden.py.
FlowNet2C/CS/CSS/FlowNet2: the GRAPHS are pinned the same way (round 4) — the imported reference
classes run with the three CUDA-only operators injected at their `_ext` FFI boundary
(make_golden.inject_restated_cuda_ops), and these functions reproduce them with max abs 0.0; the
three operators themselves stay "restated, parity unpinned" (CUDA-only in the reference).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import ops_ref

LEAK = 0.1


def _conv(sd, p, x, k, stride):
    """submodules.conv: Conv2d(+BN)+LeakyReLU(0.1); batchNorm variants detected from the keys."""
    y = F.conv2d(x, sd[p + ".0.weight"], sd.get(p + ".0.bias"), stride=stride, padding=(k - 1) // 2)
    if p + ".1.running_mean" in sd:
        y = F.batch_norm(y, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"],
                         training=False, eps=1e-5)
    return F.leaky_relu(y, LEAK)


def _deconv(sd, p, x):
    return F.leaky_relu(F.conv_transpose2d(x, sd[p + ".0.weight"], sd.get(p + ".0.bias"), stride=2, padding=1), LEAK)


def _predict(sd, p, x):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=1, padding=1)


def _upflow(sd, p, x):
    return F.conv_transpose2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=2, padding=1)


def _decoder(sd, pre, conv6, conv5, conv4, conv3, conv2):
    flow6 = _predict(sd, pre + "predict_flow6", conv6)
    concat5 = torch.cat((conv5, _deconv(sd, pre + "deconv5", conv6), _upflow(sd, pre + "upsampled_flow6_to_5", flow6)), 1)
    flow5 = _predict(sd, pre + "predict_flow5", concat5)
    concat4 = torch.cat((conv4, _deconv(sd, pre + "deconv4", concat5), _upflow(sd, pre + "upsampled_flow5_to_4", flow5)), 1)
    flow4 = _predict(sd, pre + "predict_flow4", concat4)
    concat3 = torch.cat((conv3, _deconv(sd, pre + "deconv3", concat4), _upflow(sd, pre + "upsampled_flow4_to_3", flow4)), 1)
    flow3 = _predict(sd, pre + "predict_flow3", concat3)
    concat2 = torch.cat((conv2, _deconv(sd, pre + "deconv2", concat3), _upflow(sd, pre + "upsampled_flow3_to_2", flow3)), 1)
    return _predict(sd, pre + "predict_flow2", concat2)


def flownets_trunk(sd, x, pre=""):
    c1 = _conv(sd, pre + "conv1", x, 7, 2)
    c2 = _conv(sd, pre + "conv2", c1, 5, 2)
    c3 = _conv(sd, pre + "conv3_1", _conv(sd, pre + "conv3", c2, 5, 2), 3, 1)
    c4 = _conv(sd, pre + "conv4_1", _conv(sd, pre + "conv4", c3, 3, 2), 3, 1)
    c5 = _conv(sd, pre + "conv5_1", _conv(sd, pre + "conv5", c4, 3, 2), 3, 1)
    c6 = _conv(sd, pre + "conv6_1", _conv(sd, pre + "conv6", c5, 3, 2), 3, 1)
    return _decoder(sd, pre, c6, c5, c4, c3, c2)


def _corr(a, b):
    out = ops_ref.correlation_c(a.numpy(), b.numpy(), pad_size=20, kernel_size=1, max_displacement=20, stride1=1,
                                stride2=2)
    return torch.from_numpy(out)


def flownetc_trunk(sd, x1, x2, pre="", return_features=False):
    c1a = _conv(sd, pre + "conv1", x1, 7, 2)
    c2a = _conv(sd, pre + "conv2", c1a, 5, 2)
    c3a = _conv(sd, pre + "conv3", c2a, 5, 2)
    c3b = _conv(sd, pre + "conv3", _conv(sd, pre + "conv2", _conv(sd, pre + "conv1", x2, 7, 2), 5, 2), 5, 2)
    corr = F.leaky_relu(_corr(c3a, c3b), LEAK)
    redir = _conv(sd, pre + "conv_redir", c3a, 1, 1)
    c31 = _conv(sd, pre + "conv3_1", torch.cat((redir, corr), 1), 3, 1)
    c4 = _conv(sd, pre + "conv4_1", _conv(sd, pre + "conv4", c31, 3, 2), 3, 1)
    c5 = _conv(sd, pre + "conv5_1", _conv(sd, pre + "conv5", c4, 3, 2), 3, 1)
    c6 = _conv(sd, pre + "conv6_1", _conv(sd, pre + "conv6", c5, 3, 2), 3, 1)
    flow2 = _decoder(sd, pre, c6, c5, c4, c31, c2a)
    if return_features:
        return flow2, {"conv3a": c3a, "conv3b": c3b, "corr": corr}
    return flow2


def _normalise(inputs, rgb_max):
    rgb_mean = inputs.contiguous().view(inputs.size()[:2] + (-1,)).mean(dim=-1).view(inputs.size()[:2] + (1, 1, 1))
    return (inputs - rgb_mean) / rgb_max


def _up4(flow2, div_flow):
    return F.interpolate(flow2 * div_flow, scale_factor=4, mode="bilinear", align_corners=False)


def _f32(sd):
    return {k: v.float() for k, v in sd.items() if v.is_floating_point()}


@torch.no_grad()
def flownet2s_forward(sd, inputs, rgb_max=255.0, div_flow=20.0):
    sd = _f32(sd)
    x = _normalise(inputs.float(), rgb_max)
    x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
    return _up4(flownets_trunk(sd, x), div_flow)


@torch.no_grad()
def flownet2c_forward(sd, inputs, rgb_max=255.0, div_flow=20.0):
    sd = _f32(sd)
    x = _normalise(inputs.float(), rgb_max)
    return _up4(flownetc_trunk(sd, x[:, :, 0], x[:, :, 1]), div_flow)


@torch.no_grad()
def flownet2cs_forward(sd, inputs, rgb_max=255.0, div_flow=20.0, return_parts=False):
    sd = _f32(sd)
    x = _normalise(inputs.float(), rgb_max)
    x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
    flowc = _up4(flownetc_trunk(sd, x[:, 0:3], x[:, 3:], pre="flownetc."), div_flow)
    warped = torch.from_numpy(ops_ref.resample2d_c(x[:, 3:].contiguous().numpy(), flowc.numpy()))
    diff = x[:, :3] - warped
    norm = torch.from_numpy(ops_ref.channelnorm_c(diff.numpy()))
    concat1 = torch.cat((x, warped, flowc / div_flow, norm), dim=1)
    out = _up4(flownets_trunk(sd, concat1, pre="flownets_1."), div_flow)
    if return_parts:
        return out, {"flowc": flowc, "concat1": concat1}
    return out


def _iconv(sd, p, x):
    """submodules.i_conv (submodules.py:20-29): Conv2d(+BN), no activation."""
    y = F.conv2d(x, sd[p + ".0.weight"], sd.get(p + ".0.bias"), stride=1, padding=1)
    if p + ".1.running_mean" in sd:
        y = F.batch_norm(y, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"], sd[p + ".1.bias"],
                         training=False, eps=1e-5)
    return y


def flownetsd_trunk(sd, x, pre=""):
    """FlowNetSD.forward (FlowNetSD.py:68-106) -> flow2."""
    c0 = _conv(sd, pre + "conv0", x, 3, 1)
    c1 = _conv(sd, pre + "conv1_1", _conv(sd, pre + "conv1", c0, 3, 2), 3, 1)
    c2 = _conv(sd, pre + "conv2_1", _conv(sd, pre + "conv2", c1, 3, 2), 3, 1)
    c3 = _conv(sd, pre + "conv3_1", _conv(sd, pre + "conv3", c2, 3, 2), 3, 1)
    c4 = _conv(sd, pre + "conv4_1", _conv(sd, pre + "conv4", c3, 3, 2), 3, 1)
    c5 = _conv(sd, pre + "conv5_1", _conv(sd, pre + "conv5", c4, 3, 2), 3, 1)
    c6 = _conv(sd, pre + "conv6_1", _conv(sd, pre + "conv6", c5, 3, 2), 3, 1)
    flow6 = _predict(sd, pre + "predict_flow6", c6)
    concat5 = torch.cat((c5, _deconv(sd, pre + "deconv5", c6), _upflow(sd, pre + "upsampled_flow6_to_5", flow6)), 1)
    flow5 = _predict(sd, pre + "predict_flow5", _iconv(sd, pre + "inter_conv5", concat5))
    concat4 = torch.cat((c4, _deconv(sd, pre + "deconv4", concat5), _upflow(sd, pre + "upsampled_flow5_to_4", flow5)), 1)
    flow4 = _predict(sd, pre + "predict_flow4", _iconv(sd, pre + "inter_conv4", concat4))
    concat3 = torch.cat((c3, _deconv(sd, pre + "deconv3", concat4), _upflow(sd, pre + "upsampled_flow4_to_3", flow4)), 1)
    flow3 = _predict(sd, pre + "predict_flow3", _iconv(sd, pre + "inter_conv3", concat3))
    concat2 = torch.cat((c2, _deconv(sd, pre + "deconv2", concat3), _upflow(sd, pre + "upsampled_flow3_to_2", flow3)), 1)
    return _predict(sd, pre + "predict_flow2", _iconv(sd, pre + "inter_conv2", concat2))


def flownetfusion_trunk(sd, x, pre=""):
    """FlowNetFusion.forward (FlowNetFusion.py:48-66) -> flow0 at input resolution."""
    c0 = _conv(sd, pre + "conv0", x, 3, 1)
    c1 = _conv(sd, pre + "conv1_1", _conv(sd, pre + "conv1", c0, 3, 2), 3, 1)
    c2 = _conv(sd, pre + "conv2_1", _conv(sd, pre + "conv2", c1, 3, 2), 3, 1)
    flow2 = _predict(sd, pre + "predict_flow2", c2)
    concat1 = torch.cat((c1, _deconv(sd, pre + "deconv1", c2), _upflow(sd, pre + "upsampled_flow2_to_1", flow2)), 1)
    flow1 = _predict(sd, pre + "predict_flow1", _iconv(sd, pre + "inter_conv1", concat1))
    concat0 = torch.cat((c0, _deconv(sd, pre + "deconv0", concat1), _upflow(sd, pre + "upsampled_flow1_to_0", flow1)), 1)
    return _predict(sd, pre + "predict_flow0", _iconv(sd, pre + "inter_conv0", concat0))


def _up4_nearest(x):
    return F.interpolate(x, scale_factor=4, mode="nearest")


def _warp_stage(x, flow, div_flow):
    """(x, warp(img1, flow), flow/div_flow, |img0 - warp|): models.py:124-131."""
    warped = torch.from_numpy(ops_ref.resample2d_c(x[:, 3:].contiguous().numpy(), flow.contiguous().numpy()))
    norm = torch.from_numpy(ops_ref.channelnorm_c((x[:, :3] - warped).contiguous().numpy()))
    return torch.cat((x, warped, flow / div_flow, norm), dim=1)


def _css_flow2(sd, x, div_flow):
    flowc = _up4(flownetc_trunk(sd, x[:, 0:3], x[:, 3:], pre="flownetc."), div_flow)
    flows1 = _up4(flownets_trunk(sd, _warp_stage(x, flowc, div_flow), pre="flownets_1."), div_flow)
    return flownets_trunk(sd, _warp_stage(x, flows1, div_flow), pre="flownets_2.")


@torch.no_grad()
def flownet2sd_forward(sd, inputs, rgb_max=255.0, div_flow=20.0):
    sd = _f32(sd)
    x = _normalise(inputs.float(), rgb_max)
    x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
    return _up4(flownetsd_trunk(sd, x), div_flow)


@torch.no_grad()
def flownet2css_forward(sd, inputs, rgb_max=255.0, div_flow=20.0):
    sd = _f32(sd)
    x = _normalise(inputs.float(), rgb_max)
    x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
    return _up4_nearest(_css_flow2(sd, x, div_flow) * div_flow)


@torch.no_grad()
def flownet2_forward(sd, inputs, rgb_max=255.0, div_flow=20.0, return_parts=False):
    sd = _f32(sd)
    x = _normalise(inputs.float(), rgb_max)
    x = torch.cat((x[:, :, 0], x[:, :, 1]), dim=1)
    img0, img1 = x[:, :3], x[:, 3:].contiguous()
    flows2 = _up4_nearest(_css_flow2(sd, x, div_flow) * div_flow)                      # models.py:144
    flowsd = _up4_nearest(flownetsd_trunk(sd, x, pre="flownets_d.") / div_flow)        # models.py:158 (divides)
    chn = lambda t: torch.from_numpy(ops_ref.channelnorm_c(t.contiguous().numpy()))
    warp = lambda f: torch.from_numpy(ops_ref.resample2d_c(img1.numpy(), f.contiguous().numpy()))
    concat3 = torch.cat((img0, flowsd, flows2, chn(flowsd), chn(flows2), chn(img0 - warp(flowsd)), chn(img0 - warp(flows2))), 1)
    out = flownetfusion_trunk(sd, concat3, pre="flownetfusion.")
    if return_parts:
        return out, {"concat3": concat3, "flows2": flows2, "flowsd": flowsd}
    return out


def epe(a: torch.Tensor, b: torch.Tensor) -> float:
    """End-point error, lib/flownet/model/losses.py:11-12."""
    return torch.norm(a - b, p=2, dim=1).mean().item()
