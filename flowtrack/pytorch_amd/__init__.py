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
# The switch only takes effect if no HIP call has happened yet in this process: what was found at import time is kept in
# KERNARG_STATE, logged at debug level, and bench.py prints it in its JSON line so that results stay comparable
# (import this package before the first torch.cuda call).
import logging as _logging
import sys as _sys

_preset = _os.environ.get("HIP_FORCE_DEV_KERNARG")
_torch = _sys.modules.get("torch")
_hip_up = bool(_torch is not None and _torch.cuda.is_initialized())
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
#: value = what the HIP runtime will read / has read; source = who set it; effective = False when the runtime was already initialised
#: before this package could set the variable (the setting is then ignored by this process)
KERNARG_STATE = {"value": _os.environ["HIP_FORCE_DEV_KERNARG"], "source": "environment" if _preset is not None else "package default",
                 "effective": not (_hip_up and _preset is None)}
_logging.getLogger(__name__).debug("HIP_FORCE_DEV_KERNARG=%s (%s)%s", KERNARG_STATE["value"], KERNARG_STATE["source"],
                                   "" if KERNARG_STATE["effective"] else " — HIP was initialised before the import: NOT applied")

from ._lib import FlowtrackHipError, LIB_PATH  # noqa: E402,F401

__version__ = "0.1.0"
