"""Minimal stand-in for `numba`, used ONLY by oracle/capture/capture.py.

`njit` becomes the identity decorator, so the reference's kernel bodies run
as ordinary Python/NumPy float64 code: same arithmetic, same order, slower.
"""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn
    return wrap
