from . import flow_utils, net_utils, tracker  # noqa: F401
from .flow_utils import box_propagation, nms  # noqa: F401
from .net_utils import GroupPoseRunner, PoseRunner, detect, flow_est, pose_est, pose_est_frames  # noqa: F401
from .tracker import FlowTracker  # noqa: F401
