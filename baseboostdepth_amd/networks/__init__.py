"""networks.* API of the reference (networks/__init__.py): same class names, constructor
arguments, forward signatures and state-dict keys, as PyTorch-ROCm modules (MIOpen convs)."""
from .resnet_encoder import ResnetEncoder
from .depth_decoder import DepthDecoder
from .pose_decoder import PoseDecoder

__all__ = ["ResnetEncoder", "DepthDecoder", "PoseDecoder"]
