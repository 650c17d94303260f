"""Minimal stand-in for the `mesa` package (Mesa 2.1-era API) - see ../README.md."""
import random as _random


class Agent:
    def __init__(self, unique_id, model):
        self.unique_id = unique_id
        self.model = model
        self.pos = None

    def step(self):  # pragma: no cover
        pass


class Model:
    def __new__(cls, *args, **kwargs):
        obj = object.__new__(cls)
        obj._seed = kwargs.get("seed")
        if obj._seed is None:
            obj._seed = _random.random()
        obj.random = _random.Random(obj._seed)
        return obj

    def __init__(self, *args, **kwargs):
        self.running = True
        self.schedule = None
        self.current_id = 0

    def step(self):  # pragma: no cover
        pass


from . import space, time  # noqa: E402,F401
