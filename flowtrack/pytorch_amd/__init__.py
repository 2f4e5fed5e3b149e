"""flowtrack.pytorch_amd — MI355X-native inference path for FlowTrack's two dense-CNN hot paths.

    from flowtrack.pytorch_amd.pose import models as pose_models      # models.deconv(...)
    from flowtrack.pytorch_amd.flownet import models as flow_models   # models.FlowNet2S(args) ...

Compute runs in libflowtrack_hip.so (hand-written HIP for gfx950, C ABI in include/flowtrack_hip.h);
this package is the Python host that mirrors the reference's model-factory / state_dict surface.
"""
from ._lib import FlowtrackHipError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
