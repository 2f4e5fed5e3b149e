"""Parameter holders that give the HIP models the reference's `state_dict()` contract.

The reference builds its graphs from torch.nn layers, so its checkpoints are keyed by those layers'
attribute paths (`layer1.0.conv1.weight`, `deconv.1.running_var`, `conv3_1.0.bias`, ... — SURVEY
Appendix B).  The HIP path never executes torch.nn layers; these holders only *own tensors under
the same names, shapes and dtypes* so `load_state_dict` / `state_dict` / `.cuda()` / `.half()`
behave exactly like the reference's modules.  Calling one is an error by construction.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} only stores parameters; the forward pass runs in libflowtrack_hip.so")


class ConvParams(_Holder):
    """Stands where nn.Conv2d sits in the reference (weight [Cout, Cin, k, k], optional bias)."""

    def __init__(self, cin: int, cout: int, kernel_size: int, stride: int = 1, padding: int = 0, bias: bool = True):
        super().__init__()
        self.cin, self.cout, self.kernel_size, self.stride, self.padding = cin, cout, kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(cout, cin, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        _default_conv_init(self.weight, self.bias, cin * kernel_size * kernel_size)

    def extra_repr(self):
        return f"{self.cin}, {self.cout}, k={self.kernel_size}, s={self.stride}, p={self.padding}, bias={self.bias is not None}"


class ConvTransposeParams(_Holder):
    """Stands where nn.ConvTranspose2d(cin, cout, 4, 2, 1) sits (weight [Cin, Cout, 4, 4])."""

    def __init__(self, cin: int, cout: int, bias: bool = True):
        super().__init__()
        self.cin, self.cout, self.kernel_size, self.stride, self.padding = cin, cout, 4, 2, 1
        self.weight = nn.Parameter(torch.empty(cin, cout, 4, 4))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        _default_conv_init(self.weight, self.bias, cout * 16)  # torch's fan_in for ConvTranspose weight layout

    def extra_repr(self):
        return f"{self.cin}, {self.cout}, k=4, s=2, p=1, bias={self.bias is not None}"


class BatchNormParams(_Holder):
    """Stands where nn.BatchNorm2d sits: weight, bias, running_mean, running_var, num_batches_tracked."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        # checkpoints written by torch 0.4.0 have no num_batches_tracked (SURVEY §8(c) drift item 4)
        key = prefix + "num_batches_tracked"
        if key not in state_dict:
            state_dict[key] = torch.tensor(0, dtype=torch.long)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def as_dict(self) -> dict:
        return {"weight": self.weight, "bias": self.bias, "running_mean": self.running_mean,
                "running_var": self.running_var, "eps": self.eps}


class ActMarker(_Holder):
    """Parameter-free placeholder keeping nn.Sequential indices aligned with the reference
    (e.g. pose_deconv.py:19-28 has ReLU at deconv.2/.5/.8)."""

    def __init__(self, kind: str, slope: float = 0.0):
        super().__init__()
        self.kind, self.slope = kind, slope

    def extra_repr(self):
        return self.kind if self.kind != "leaky" else f"leaky({self.slope})"


def _default_conv_init(weight: torch.Tensor, bias, fan_in: int) -> None:
    # torch.nn's default reset_parameters(): kaiming_uniform(a=sqrt(5)) and U(-1/sqrt(fan_in), ..)
    with torch.no_grad():
        bound = 1.0 / math.sqrt(fan_in)
        weight.uniform_(-bound, bound)
        if bias is not None:
            bias.uniform_(-bound, bound)
