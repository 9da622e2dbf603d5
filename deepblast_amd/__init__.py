"""deepblast_amd -- MI355X-native differentiable soft-DP alignment engine.

Drop-in replacement for the DP operators of flatironinstitute/deepblast
(deepblast/nw_cuda.py, deepblast/sw_cuda.py): hand-written HIP kernels for gfx950
behind a C ABI (include/sdp.h), called through ctypes with raw PyTorch-ROCm pointers.
"""
from .nw import NeedlemanWunschDecoder, NeedlemanWunschFunction, NeedlemanWunschFunctionBackward
from .sw import SmithWatermanDecoder, SmithWatermanFunction, SmithWatermanFunctionBackward

__all__ = [
    "NeedlemanWunschDecoder", "NeedlemanWunschFunction", "NeedlemanWunschFunctionBackward",
    "SmithWatermanDecoder", "SmithWatermanFunction", "SmithWatermanFunctionBackward",
]
__version__ = "0.1.0"
