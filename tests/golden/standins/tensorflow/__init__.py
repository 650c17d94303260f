"""Permissive `tensorflow` stub - only so `import tensorflow as tf` succeeds."""
import sys
import types


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Any()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Any()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Mod(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Any()


def _install(name):
    m = _Mod(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


for _n in ("tensorflow.keras", "tensorflow.keras.layers", "tensorflow.keras.optimizers",
           "tensorflow.keras.models", "tensorflow.keras.losses", "tensorflow.config",
           "tensorflow.keras.initializers", "tensorflow.keras.backend"):
    _install(_n)


def __getattr__(name):
    if name.startswith("__") and name.endswith("__"):
        raise AttributeError(name)
    return _Any()
