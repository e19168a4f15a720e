"""networks.* API of the reference (same class names, constructor arguments, forward signatures and
state-dict keys), as PyTorch-ROCm modules: the conv stacks run on MIOpen."""
from .decoders import DepthDecoder, PoseDecoder
from .encoder import ResnetEncoder

__all__ = ["ResnetEncoder", "DepthDecoder", "PoseDecoder"]
