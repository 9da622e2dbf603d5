class NumbaPerformanceWarning(Warning):
    """nw_cuda.py:6 imports the name."""
