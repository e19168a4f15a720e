"""networksvit.* API of the reference (MonoViT: MPViT encoder + HR-Depth decoder; BASELINE configs[4]).
Same names, constructor arguments and state-dict keys as /root/reference/networksvit; PyTorch-ROCm
modules (MIOpen / rocBLAS) feeding the same HIP photometric-loss kernels as the ResNet path."""
from .hr_decoder import DepthDecoder
from .mpvit import MPViT, mpvit_tiny, mpvit_xsmall, mpvit_small, mpvit_base

__all__ = ["DepthDecoder", "MPViT", "mpvit_tiny", "mpvit_xsmall", "mpvit_small", "mpvit_base"]
