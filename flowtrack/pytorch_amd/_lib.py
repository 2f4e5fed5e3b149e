"""ctypes binding of libflowtrack_hip.so (the C ABI declared in include/flowtrack_hip.h).

This is the Python-side FFI stub that takes the place of the reference's cffi `_ext` modules
(lib/flownet/networks/*/_ext/*/__init__.py) and of the torch.nn -> cuDNN dispatch under
lib/pose/models.  There is no CPU or PyTorch fallback: if the shared library is missing or a call
returns a non-zero status, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FT_LIB_PATH") or os.path.join(HERE, "libflowtrack_hip.so")  # FT_LIB_PATH: developer A/B builds

FT_OK = 0
FT_ERR_UNSUPPORTED = 2   # ft_status: valid but not implemented for this combination
FT_F16, FT_F32 = 0, 1
FT_ACT_NONE, FT_ACT_RELU, FT_ACT_LEAKY = 0, 1, 2
FT_LAYOUT_NHWC, FT_LAYOUT_NCHW_F32 = 0, 1
FT_RGB_MEAN_SPLITS = 64


class FlowtrackHipError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    """Mirror of `ft_conv_desc` (include/flowtrack_hip.h)."""

    _fields_ = [
        ("dtype", c_int), ("N", c_int), ("Hi", c_int), ("Wi", c_int), ("Cin", c_int),
        ("x_cstride", c_int), ("x_coff", c_int), ("Cout", c_int), ("kh", c_int), ("kw", c_int),
        ("stride", c_int), ("pad", c_int), ("transposed", c_int), ("Ho", c_int), ("Wo", c_int),
        ("y_cstride", c_int), ("y_coff", c_int), ("out_layout", c_int), ("has_residual", c_int),
        ("res_cstride", c_int), ("res_coff", c_int), ("act", c_int), ("slope", c_float),
        ("x_lpad", c_int), ("x_wpitch", c_int), ("tile_hint", c_int),
        ("x2_cin", c_int), ("x2_hi", c_int), ("x2_wi", c_int), ("x2_cstride", c_int), ("x2_coff", c_int), ("x2_stride", c_int),
        ("tail_cout", c_int), ("pool", c_int), ("shift_nstride", c_int), ("x_nchw_f32", c_int),
    ]


class BottleneckDesc(ctypes.Structure):
    """Mirror of `ft_bottleneck_desc`."""

    _fields_ = [("dtype", c_int), ("N", c_int), ("H", c_int), ("W", c_int), ("C", c_int), ("P", c_int),
                ("x_cstride", c_int), ("x_coff", c_int), ("y_cstride", c_int), ("y_coff", c_int), ("head_only", c_int), ("projection", c_int),
                ("stride", c_int), ("folded", c_int)]


class ConvGeometry(ctypes.Structure):
    """Mirror of `ft_conv_geometry`."""

    _fields_ = [("nphases", c_int), ("ntaps", c_int), ("cin_pad", c_int), ("cout_pad", c_int), ("kpad", c_int),
                ("run_taps", c_int), ("run_cpad", c_int), ("cin2_pad", c_int)]

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


_PROTOTYPES = {
    # name: (restype, argtypes)
    "ft_version": (c_int, []),
    "ft_status_string": (c_char_p, [c_int]),
    "ft_last_hip_error": (c_char_p, []),
    "ft_device_info": (c_int, [c_int, c_char_p, c_int, POINTER(c_int), POINTER(c_uint64)]),
    "ft_graph_begin_capture": (c_int, [c_void_p]),
    "ft_graph_end_capture": (c_int, [c_void_p, POINTER(c_void_p)]),
    "ft_graph_launch": (c_int, [c_void_p, c_void_p]),
    "ft_graph_destroy": (c_int, [c_void_p]),
    "ft_event_create": (c_int, [POINTER(c_void_p)]),
    "ft_event_record": (c_int, [c_void_p, c_void_p]),
    "ft_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "ft_event_synchronize": (c_int, [c_void_p]),
    "ft_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "ft_event_destroy": (c_int, [c_void_p]),
    "ft_stream_synchronize": (c_int, [c_void_p]),
    "ft_memcpy_async": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "ft_conv_pack_geometry": (c_int, [POINTER(ConvDesc), POINTER(ConvGeometry)]),
    "ft_conv_tile_candidates": (c_int, [POINTER(ConvDesc), POINTER(c_int), c_int]),
    "ft_conv_workspace_bytes": (ctypes.c_size_t, [POINTER(ConvDesc)]),
    "ft_conv2d_fwd_ws": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 ctypes.c_size_t, c_void_p]),
    "ft_conv_tap_source": (c_int, [POINTER(ConvDesc), c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "ft_conv2d_fwd": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p]),
    "ft_conv_flops": (c_double, [POINTER(ConvDesc)]),
    "ft_conv_direct_supported": (c_int, [POINTER(ConvDesc)]),
    "ft_conv_direct_stream_id": (c_int, [POINTER(ConvDesc)]),
    "ft_conv_direct_weight_bytes": (ctypes.c_longlong, [POINTER(ConvDesc)]),
    "ft_conv_direct_pack": (c_int, [POINTER(ConvDesc), c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ft_conv_direct_fwd": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ft_bottleneck_supported": (c_int, [POINTER(BottleneckDesc)]),
    "ft_bottleneck_fwd": (c_int, [POINTER(BottleneckDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ft_bottleneck_flops": (c_double, [POINTER(BottleneckDesc)]),
    "ft_bottleneck_rstat_supported": (c_int, [POINTER(BottleneckDesc)]),
    "ft_bottleneck_rstat_weight_bytes": (ctypes.c_longlong, []),
    "ft_bottleneck_rstat_fwd": (c_int, [POINTER(BottleneckDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "ft_bottleneck_stream_supported": (c_int, [POINTER(BottleneckDesc)]),
    "ft_bottleneck_stream_folds": (c_int, [POINTER(BottleneckDesc)]),
    "ft_bottleneck_stream_weight_bytes": (ctypes.c_longlong, [POINTER(BottleneckDesc)]),
    "ft_bottleneck_stream_pack": (c_int, [POINTER(BottleneckDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ft_bottleneck_stream_fwd": (c_int, [POINTER(BottleneckDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ft_pack_nchw_to_nhwc": (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "ft_unpack_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    "ft_maxpool3x3s2_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ft_bn_batch_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ft_heatmap_max_preds": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "ft_heatmap_keypoint_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ft_heatmap_min_margin": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ft_heatmap_argmax_screen": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ft_flow_rgb_mean": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ft_flow_pack_pair": (c_int, [c_void_p, c_void_p, c_float, c_void_p] + [c_int] * 7 + [c_void_p]),
    "ft_flow_mean_pack_pair_state_words": (ctypes.c_longlong, [c_int, c_int, c_int]),
    "ft_flow_pack_pair_sums_chunks": (ctypes.c_longlong, [c_int]),
    "ft_conv_shift_nstride_supported": (c_int, [POINTER(ConvDesc)]),
    "ft_flow_pack_pair_sums": (c_int, [c_void_p, c_float, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p]),
    "ft_flow_mean_fold": (c_int, [c_void_p, c_float, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    "ft_flow_mean_pack_pair": (c_int, [c_void_p, c_float, c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_int] * 4
                               + [c_void_p, c_void_p, c_void_p]),
    "ft_upsample_bilinear4x": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ft_correlation_out_shape": (c_int, [c_int] * 8 + [POINTER(c_int)] * 3),
    "ft_correlation_fwd": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "ft_correlation_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_float, c_int, c_void_p]),
    "ft_resample2d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ft_channelnorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ft_upsample_nearest4x": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ft_flow_fusion_concat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    "ft_crop_affine_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                   c_void_p, c_void_p]),
    "ft_flow_warp_concat": (c_int, [c_void_p, c_void_p, c_float, c_void_p] + [c_int] * 8 + [c_void_p]),
    "ft_gather_flagged_rows": (c_int, [c_void_p, c_int, c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_void_p]),
    "ft_bottleneck_cluster_supported": (c_int, [c_void_p]),
    "ft_bottleneck_cluster_workspace_bytes": (ctypes.c_longlong, [c_void_p]),
    "ft_bottleneck_cluster_status_offset": (ctypes.c_longlong, [c_void_p]),
    "ft_bottleneck_cluster_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ft_crop_affine_cv2_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                       c_void_p, c_void_p, c_void_p]),
}

# declared only under FT_EXPERIMENTAL in the header: measured alternatives the default plans do not record (tests / tools/dev reach them)
EXPERIMENTAL_SYMBOLS = ("ft_bottleneck_rstat_supported", "ft_bottleneck_rstat_weight_bytes", "ft_bottleneck_rstat_fwd", "ft_bottleneck_cluster_supported", "ft_bottleneck_cluster_workspace_bytes", "ft_bottleneck_cluster_status_offset", "ft_bottleneck_cluster_fwd")
EXPORTED_SYMBOLS = tuple(n for n in _PROTOTYPES if n not in EXPERIMENTAL_SYMBOLS)

_lib = None


def load():
    """Load libflowtrack_hip.so once; raises FlowtrackHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FlowtrackHipError(
            f"{LIB_PATH} is missing: build it with `python -m flowtrack.pytorch_amd.build` "
            "(there is no CPU / PyTorch fallback for the HIP path)")
    # torch must be imported first so that the process has ONE HIP runtime: the library's
    # DT_NEEDED libamdhip64.so.7 then resolves to the copy torch already loaded.
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in _PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != FT_OK:
        lib = load()
        msg = lib.ft_status_string(status).decode()
        detail = lib.ft_last_hip_error().decode()
        raise FlowtrackHipError(f"{what or 'libflowtrack_hip'}: {msg}" + (f" ({detail})" if detail else ""))


def dtype_code(torch_dtype) -> int:
    import torch

    if torch_dtype == torch.float16:
        return FT_F16
    if torch_dtype == torch.float32:
        return FT_F32
    raise FlowtrackHipError(f"unsupported dtype {torch_dtype}; the HIP path computes in fp16 or fp32")
