"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional torch-CPU fp32 restatement of the pose network, driven by a reference-keyed state_dict.
Follows DeconvResnet.forward (lib/pose/models/pose_deconv.py:32-46), ResNet (resnet.py:15-36) and
Bottleneck.forward (blocks.py:105-120); BatchNorm in eval mode (tools/pose/main.py:259).
Every op is a separate torch.nn.functional call exactly as the reference executes them on CPU.
Pinned against the imported reference by tests/golden/make_golden.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

RESNET_LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=False, eps=1e-5)


def _bottleneck(sd, p, x, stride):
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1)))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    if p + ".downsample.0.weight" in sd:
        residual = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
    else:
        residual = x
    return F.relu(out + residual)


STAGES = ("input", "stem", "layer1", "layer2", "layer3", "layer4", "deconv.0", "deconv.3", "deconv.6")


@torch.no_grad()
def pose_forward(sd, x, depth=50, return_features=False, start="input"):
    """sd: state_dict of deconv('resnet<depth>', K) (fp32 CPU tensors); x [B,3,H,W] -> [B,K,H/4,W/4].
    start (tests/error_budget.py): `x` is the activation BEHIND that stage (one of STAGES, NCHW) and the forward continues from
    there — the same calls in the same order, only the earlier ones skipped."""
    sd = {k: v.float() for k, v in sd.items() if v.is_floating_point()}
    x = x.float()
    feats = {}
    at = STAGES.index(start)
    if at < 1:
        x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3)))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats["stem"] = x
    for li, nblocks in enumerate(RESNET_LAYERS[depth], start=1):
        if at < 1 + li:
            for bi in range(nblocks):
                stride = 2 if (bi == 0 and li > 1) else 1
                x = _bottleneck(sd, f"layer{li}.{bi}", x, stride)
        feats[f"layer{li}"] = x
    for j, i in enumerate((0, 3, 6)):
        if at < 6 + j:
            x = F.conv_transpose2d(x, sd[f"deconv.{i}.weight"], stride=2, padding=1)
            x = F.relu(_bn(sd, f"deconv.{i + 1}", x))
        feats[f"deconv.{i}"] = x
    feats["deconv"] = x
    x = F.conv2d(x, sd["heatmap.weight"], sd["heatmap.bias"])
    return (x, feats) if return_features else x
