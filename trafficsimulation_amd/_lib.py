"""Loader for the HIP engine (csrc/libtrafficsim_hip.so).  There is NO CPU fallback: if the
library is missing or no gfx950 device is visible, the product fails loudly."""
from __future__ import annotations

import ctypes
import os

from ._capi import CApi

HERE = os.path.dirname(os.path.abspath(__file__))
# (TS_HIP_LIB: another build of the same engine - the test build with the small LDS heap, an experiment - in place of the default)
LIB_PATH = os.environ.get("TS_HIP_LIB") or os.path.join(HERE, "csrc", "libtrafficsim_hip.so")
_lib = None


class EngineUnavailable(RuntimeError):
    pass


def load_library() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineUnavailable(
                f"{LIB_PATH} is missing - build it with `make -C trafficsimulation_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def new_engine() -> CApi:
    """A fresh, un-created engine handle bound to the HIP library (prefix ts_)."""
    return CApi(load_library(), "ts_")
