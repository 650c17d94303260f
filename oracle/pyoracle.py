"""Loader for the CPU oracle (oracle/libtso.so).  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg - never by trafficsimulation_amd."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libtso.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "tso.cpp")
    hdr = os.path.join(HERE, "..", "include", "trafficsim.h")
    stale = (not os.path.exists(LIB)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB) for s in (src, hdr))
    if force or stale:
        subprocess.run(["make", "-C", HERE, "-s"] + (["-B"] if force else []), check=True)
    return LIB


def load():
    """-> trafficsimulation_amd._capi.CApi bound to the oracle (prefix tso_)."""
    from trafficsimulation_amd._capi import CApi
    build()
    return CApi(ctypes.CDLL(LIB), "tso_")
