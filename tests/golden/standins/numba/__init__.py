"""`numba` stand-in: njit is the identity decorator - see ../README.md."""


def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn
    return deco


jit = njit


class _T:
    def __call__(self, *a, **k):
        return self

    def __getitem__(self, item):
        return self


int8 = int16 = int32 = int64 = uint8 = float32 = float64 = boolean = _T()


def prange(*a):
    return range(*a)


from . import core, typed  # noqa: E402,F401
