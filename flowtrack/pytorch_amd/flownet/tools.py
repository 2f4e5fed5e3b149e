"""Argument reflection + flow file I/O used by the flownet entry point.

module_to_dict / add_arguments_for_module / kwargs_from_args mirror lib/flownet/utils/tools.py:19-22,55-86
(class constructor kwargs become `--model_<kw>` flags).  write_flow / read_flow implement the Middlebury .flo
format as lib/flownet/utils/flowlib.py:132-148 and flow_utils.py:5-55 do (magic 202021.25, int32 w, h,
interleaved fp32 u, v).  flow_to_image is the usual Middlebury colour wheel (flowlib.py:243-282), numpy only.
"""
from __future__ import annotations

import inspect
from inspect import isclass

import numpy as np

TAG_FLOAT = 202021.25


def module_to_dict(module, exclude=()):
    return dict([(x, getattr(module, x)) for x in dir(module)
                 if isclass(getattr(module, x)) and x not in exclude and getattr(module, x) not in exclude])


def add_arguments_for_module(parser, module, argument_for_class, default, skip_params=(), parameter_defaults=None,
                             choices=None):
    parameter_defaults = parameter_defaults or {}
    group = parser.add_argument_group(argument_for_class.capitalize())
    module_dict = module_to_dict(module)
    group.add_argument('--' + argument_for_class, type=str, default=default, choices=choices or list(module_dict.keys()))
    args, _ = parser.parse_known_args()
    class_obj = module_dict[vars(args)[argument_for_class]]
    sig = inspect.signature(class_obj.__init__)
    for name, prm in sig.parameters.items():
        if name in ('self', 'args') or name in skip_params:
            continue
        cmd = '{}_{}'.format(argument_for_class, name)
        if name in parameter_defaults:
            group.add_argument('--' + cmd, type=type(parameter_defaults[name]), default=parameter_defaults[name])
        elif prm.default is not inspect.Parameter.empty:
            group.add_argument('--' + cmd, type=type(prm.default), default=prm.default)
        else:
            print("[Warning]: non-default argument '{}' detected on class '{}'. This argument cannot be modified via "
                  "the command line".format(name, class_obj.__name__))


def kwargs_from_args(args, argument_for_class):
    prefix = argument_for_class + '_'
    return {k[len(prefix):]: v for k, v in vars(args).items() if prefix in k and k != prefix + 'class'}


def write_flow(flow: np.ndarray, filename: str) -> None:
    """flow [H, W, 2] fp32 -> .flo"""
    flow = np.asarray(flow, dtype=np.float32)
    h, w = flow.shape[:2]
    with open(filename, 'wb') as f:
        np.array([TAG_FLOAT], dtype=np.float32).tofile(f)
        np.array([w, h], dtype=np.int32).tofile(f)
        flow.tofile(f)


def read_flow(filename: str) -> np.ndarray:
    with open(filename, 'rb') as f:
        magic = np.fromfile(f, np.float32, count=1)[0]
        if magic != np.float32(TAG_FLOAT):
            raise ValueError('bad .flo magic in ' + filename)
        w, h = np.fromfile(f, np.int32, count=2)
        return np.fromfile(f, np.float32, count=2 * w * h).reshape(h, w, 2)


def _color_wheel():
    RY, YG, GC, CB, BM, MR = 15, 6, 4, 11, 13, 6
    ncols = RY + YG + GC + CB + BM + MR
    wheel = np.zeros((ncols, 3))
    col = 0
    for n, (a, b, sign) in zip((RY, YG, GC, CB, BM, MR), ((0, 1, 1), (1, 0, -1), (1, 2, 1), (2, 1, -1), (2, 0, 1), (0, 2, -1))):
        ramp = np.floor(255 * np.arange(n) / n)
        wheel[col:col + n, a] = 255
        wheel[col:col + n, b] = ramp if sign > 0 else 255 - ramp
        if sign < 0:
            wheel[col:col + n, a], wheel[col:col + n, b] = 255 - ramp, 255
        col += n
    return wheel


def flow_to_image(flow: np.ndarray) -> np.ndarray:
    """[H, W, 2] -> uint8 RGB, normalised by the maximum flow magnitude."""
    u, v = flow[..., 0].astype(np.float64), flow[..., 1].astype(np.float64)
    bad = (np.abs(u) > 1e7) | (np.abs(v) > 1e7)
    u[bad] = v[bad] = 0
    rad = np.sqrt(u * u + v * v)
    maxrad = max(rad.max(), np.finfo(float).eps)
    u, v, rad = u / maxrad, v / maxrad, rad / maxrad
    wheel = _color_wheel()
    ncols = wheel.shape[0]
    fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (ncols - 1)
    k0 = np.floor(fk).astype(int)
    k1 = (k0 + 1) % ncols
    f = fk - k0
    img = np.zeros(u.shape + (3,), dtype=np.uint8)
    for i in range(3):
        col = (1 - f) * wheel[k0, i] / 255.0 + f * wheel[k1, i] / 255.0
        inside = rad <= 1
        col[inside] = 1 - rad[inside] * (1 - col[inside])
        col[~inside] *= 0.75
        img[..., i] = np.floor(255 * col * (~bad)).astype(np.uint8)
    return img
