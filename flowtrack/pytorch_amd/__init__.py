"""flowtrack.pytorch_amd — MI355X-native inference path for FlowTrack's two dense-CNN hot paths.

    from flowtrack.pytorch_amd.pose import models as pose_models      # models.deconv(...)
    from flowtrack.pytorch_amd.flownet import models as flow_models   # models.FlowNet2S(args) ...

Compute runs in libflowtrack_hip.so (hand-written HIP for gfx950, C ABI in include/flowtrack_hip.h);
this package is the Python host that mirrors the reference's model-factory / state_dict surface.
"""
import os as _os

# Kernel arguments in device memory (ROCm runtime switch, read when the HIP runtime initialises — i.e. at the first HIP call, so
# importing this package before touching the GPU is enough): the plans are graphs of 30-60 small launches, some on parallel
# branches, and the kernarg fetch is part of every launch's latency.  Measured on FlowNet2S (16 x 512x384, two interleaved runs):
# 19.28 / 19.30 -> 20.02 / 20.01 k pairs/s (+3.7 %); the ResNet-50 pose step (29 larger launches) does not move.  An explicit
# HIP_FORCE_DEV_KERNARG in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from ._lib import FlowtrackHipError, LIB_PATH  # noqa: E402,F401

__version__ = "0.1.0"
