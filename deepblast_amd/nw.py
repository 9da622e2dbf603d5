"""Needleman-Wunsch soft-DP operator on MI355X.

Drop-in for deepblast.nw_cuda (NeedlemanWunschDecoder & friends, nw_cuda.py:168-325):
same class names, call signatures, error behaviour and gradient semantics; the Numba-CUDA
kernels are replaced by the HIP engine behind include/sdp.h.
"""
from . import _dp
from ._engine import NW

NeedlemanWunschFunction, NeedlemanWunschFunctionBackward = _dp.make_functions(NW, "NeedlemanWunsch")


class NeedlemanWunschDecoder(_dp._Decoder):
    _function = NeedlemanWunschFunction
