"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Correlation / Resample2d / ChannelNorm / x4 bilinear upsample on the CPU, two ways:
  *_c   : ctypes calls into oracle/_build/libflowtrack_oracle.so (flow_ops_ref.c — a step-by-step C
          restatement of the reference CUDA kernels, incl. their padded NHWC staging and summation order)
  *_np  : an independent vectorised numpy formulation from the operator definitions
Tests require the two to agree before either is used to judge the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libflowtrack_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "flow_ops_ref.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---- correlation ------------------------------------------------------------------------------------
def correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2):
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    rc = lib().correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                     ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow))
    if rc:
        raise ValueError("empty correlation output")
    return oc.value, oh.value, ow.value


def correlation_c(in1, in2, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2):
    in1, in2 = _f32(in1), _f32(in2)
    B, C, H, W = in1.shape
    oc, oh, ow = correlation_out_shape(C, H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    out = np.zeros((B, oc, oh, ow), dtype=np.float32)
    rc = lib().correlation_fwd(_p(in1), _p(in2), _p(out), B, C, H, W, pad_size, kernel_size, max_displacement,
                               stride1, stride2)
    if rc:
        raise RuntimeError(f"correlation_fwd failed ({rc})")
    return out


def correlation_np(in1, in2, pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2):
    """Definition: mean over (window, channel) of products of in1 patches and displaced in2 patches,
    zero outside the image; float64 accumulation."""
    in1 = np.asarray(in1, dtype=np.float64)
    in2 = np.asarray(in2, dtype=np.float64)
    B, C, H, W = in1.shape
    krad = (kernel_size - 1) // 2
    drad = max_displacement // stride2
    D = 2 * drad + 1
    border = krad + max_displacement
    oh = int(np.ceil((H + 2 * pad_size - 2 * border) / stride1))
    ow = int(np.ceil((W + 2 * pad_size - 2 * border) / stride1))
    # generous zero padding so every index below is valid
    P = pad_size + border + stride1 * max(oh, ow) + 2
    p1 = np.pad(in1, ((0, 0), (0, 0), (P, P), (P, P)))
    p2 = np.pad(in2, ((0, 0), (0, 0), (P, P), (P, P)))
    out = np.zeros((B, D * D, oh, ow))
    ys = np.arange(oh) * stride1 + max_displacement + krad - pad_size + P
    xs = np.arange(ow) * stride1 + max_displacement + krad - pad_size + P
    for tj in range(-drad, drad + 1):
        for ti in range(-drad, drad + 1):
            acc = np.zeros((B, oh, ow))
            for j in range(-krad, krad + 1):
                for i in range(-krad, krad + 1):
                    a = p1[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    b = p2[:, :, (ys + j + tj * stride2)[:, None], (xs + i + ti * stride2)[None, :]]
                    acc += (a * b).sum(1)
            out[:, (tj + drad) * D + (ti + drad)] = acc / (kernel_size * kernel_size * C)
    return out.astype(np.float32)


# ---- resample2d -------------------------------------------------------------------------------------
def resample2d_c(in1, flow):
    in1, flow = _f32(in1), _f32(flow)
    B, C, H, W = in1.shape
    out = np.zeros((B, C, H, W), dtype=np.float32)
    lib().resample2d_fwd(_p(in1), _p(flow), _p(out), B, C, H, W)
    return out


def resample2d_np(in1, flow):
    in1 = np.asarray(in1, dtype=np.float64)
    flow = np.asarray(flow, dtype=np.float32)
    B, C, H, W = in1.shape
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    xf = xx[None] + flow[:, 0]
    yf = yy[None] + flow[:, 1]
    fx, fy = np.floor(xf), np.floor(yf)
    a = (xf - fx).astype(np.float64)[:, None]
    b = (yf - fy).astype(np.float64)[:, None]
    xL = np.clip(fx, 0, W - 1).astype(np.int64)
    xR = np.clip(fx + 1, 0, W - 1).astype(np.int64)
    yT = np.clip(fy, 0, H - 1).astype(np.int64)
    yB = np.clip(fy + 1, 0, H - 1).astype(np.int64)
    bi = np.arange(B)[:, None, None]
    g = lambda yi, xi: np.stack([in1[bi, c, yi, xi] for c in range(C)], axis=1)
    out = (1 - a) * (1 - b) * g(yT, xL) + a * (1 - b) * g(yT, xR) + (1 - a) * b * g(yB, xL) + a * b * g(yB, xR)
    return out.astype(np.float32)


# ---- channelnorm --------------------------------------------------------------------------------------
def channelnorm_c(in1):
    in1 = _f32(in1)
    B, C, H, W = in1.shape
    out = np.zeros((B, 1, H, W), dtype=np.float32)
    lib().channelnorm_fwd(_p(in1), _p(out), B, C, H, W)
    return out


def channelnorm_np(in1):
    return np.sqrt((np.asarray(in1, dtype=np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)


# ---- x4 bilinear upsample ---------------------------------------------------------------------------
def upsample4x_c(x, mul=1.0):
    x = _f32(x)
    N, C, h, w = x.shape
    y = np.zeros((N, C, 4 * h, 4 * w), dtype=np.float32)
    lib().upsample4x_fwd(_p(x), _p(y), N, C, h, w, ctypes.c_float(mul))
    return y


# ---- direct conv definitions (cross-check of the torch functional oracle) -------------------------------
def conv2d_direct(x, w, bias, stride, pad):
    x, w = _f32(x), _f32(w)
    N, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = np.zeros((N, Cout, Ho, Wo), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    lib().conv2d_direct(_p(x), _p(w), _p(b) if b is not None else None, _p(y), N, Cin, H, W, Cout, k, stride, pad)
    return y


def conv_transpose2d_direct(x, w, bias, stride=2, pad=1):
    x, w = _f32(x), _f32(w)
    N, Cin, H, W = x.shape
    _, Cout, k, _ = w.shape
    Ho, Wo = (H - 1) * stride - 2 * pad + k, (W - 1) * stride - 2 * pad + k
    y = np.zeros((N, Cout, Ho, Wo), dtype=np.float32)
    b = _f32(bias) if bias is not None else None
    lib().conv_transpose2d_direct(_p(x), _p(w), _p(b) if b is not None else None, _p(y), N, Cin, H, W, Cout, k,
                                  stride, pad)
    return y
