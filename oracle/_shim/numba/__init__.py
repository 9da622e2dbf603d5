"""No-op stand-in for `numba` so the reference's pure-Python recurrences import
in this container (numba is not installed).  Test infrastructure only: used by
oracle/gen_golden*.py to run /root/reference/deepblast/{nw,sw}.py -- and to import
nw_cuda.py / sw_cuda.py for their host-side traceback (the @cuda.jit kernels are
never called) -- and emit the committed fixtures under tests/golden/.  Never
imported by the product."""


def _identity(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda fn: fn


njit = jit = _identity
float32 = "float32"   # nw_cuda.py:12 names the type at import time; nothing here uses it
