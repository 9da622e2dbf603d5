"""DeepBLAST's "Smith-Waterman" soft-DP operator on MI355X.

Drop-in for deepblast.sw_cuda (SmithWatermanDecoder & friends, sw_cuda.py:168-326).  As in the
reference this is the Needleman-Wunsch recurrence with padded row 1 / column 1 skipped in the
forward and backward sweeps (sw.py:54-55,107-110) -- not a 4-state affine local aligner.
The reference CPU class is constructed with operator=None in its tests (test_sw.py:40), the GPU
class with 'softmax' (alignment.py:74); both are accepted.
"""
from . import _dp
from ._engine import SW

SmithWatermanFunction, SmithWatermanFunctionBackward = _dp.make_functions(SW, "SmithWaterman",
                                                                          allow_none_operator=True)


class SmithWatermanDecoder(_dp._Decoder):
    _function = SmithWatermanFunction
