"""The engine's host-side word pipe (csrc/mt19937.h: producer thread, tempered-word ring, checkpoints for
random.getstate(), accept bitmask for _randbelow) against CPython's own `random`, on the CPU: the header is plain C++.
A second build runs the same consumption pattern under ThreadSanitizer."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "mtpipe_check.cpp")
BUILD = os.path.join(HERE, "native", "_build")


def _build(name, extra):
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, name)
    hdr = os.path.join(HERE, "..", "trafficsimulation_amd", "csrc", "mt19937.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", *extra, "-o", out, SRC], check=True)
    return out


@pytest.fixture(scope="module")
def lib():
    L = ctypes.CDLL(_build("libmtpipe_check.so", ["-shared", "-fPIC"]))
    L.mtpipe_run.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32] + [ctypes.c_void_p] * 4
    L.mtpipe_rolls.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return L


def _state(r):
    st = r.getstate()[1]
    return np.asarray(st[:624], dtype=np.uint32), int(st[624])


@pytest.mark.parametrize("seed,warm,n,stride", [(1, 0, 5000, 1), (7, 123, 700001, 997), (11, 624, 9_500_000, 4096)])
def test_words_and_state_match_cpython(lib, seed, warm, n, stride):
    """n words after setstate (the last case runs past the 2^23-word ring and 230 checkpoints): the final words and the
    state a random.getstate() would return at that point."""
    r = random.Random(seed)
    for _ in range(warm):
        r.getrandbits(32)
    mt, idx = _state(r)
    fold, last, mt_out, idx_out = np.zeros(1, np.uint32), np.zeros(16, np.uint32), np.zeros(624, np.uint32), np.zeros(1, np.uint32)
    lib.mtpipe_run(mt.ctypes.data, idx, n, stride, fold.ctypes.data, last.ctypes.data, mt_out.ctypes.data, idx_out.ctypes.data)
    # CPython side: getrandbits(32 * k) yields k words, lowest first
    bulk = n - 16
    while bulk > 0:
        k = min(bulk, 1 << 20)
        r.getrandbits(32 * k)
        bulk -= k
    want_last = [r.getrandbits(32) for _ in range(16)]
    assert last.tolist() == want_last
    w_mt, w_idx = _state(r)
    assert int(idx_out[0]) == w_idx and np.array_equal(mt_out, w_mt)


@pytest.mark.parametrize("span", [5, 1000, 499_931, 1_000_003, (1 << 20) + 1])
def test_randbelow_and_take_match_cpython(lib, span):
    r = random.Random(span)
    mt, idx = _state(r)
    n = 3000
    values, takes = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    lib.mtpipe_rolls(mt.ctypes.data, idx, span, n, values.ctypes.data, takes.ctypes.data)
    k = span.bit_length()
    want_v, want_t = [], []
    for _ in range(n):
        t = 1
        v = r.getrandbits(k)          # random.Random._randbelow_with_getrandbits
        while v >= span:
            v = r.getrandbits(k)
            t += 1
        want_v.append(v)
        want_t.append(t)
    assert values.tolist() == want_v
    assert [t for t in takes.tolist()] == [t if t <= 64 else 0 for t in want_t]


def test_pipe_is_race_free_under_tsan():
    # target_clones resolvers run before the sanitizer's runtime is up; the header drops the clones when
    # __HIP_DEVICE_COMPILE__ is defined, which this host-only build borrows to get the plain functions
    exe = _build("mtpipe_check_tsan", ["-g", "-fsanitize=thread", "-DMTPIPE_MAIN", "-D__HIP_DEVICE_COMPILE__=1"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert p.returncode == 0 and p.stdout.startswith("ok "), p.stderr[-2000:]
    assert "WARNING: ThreadSanitizer" not in p.stderr
