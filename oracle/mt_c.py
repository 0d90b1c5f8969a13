"""oracle/mt_c.py -- ctypes wrapper of oracle/mt19937_legacy.c (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmt19937_legacy.so")


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "mt19937_legacy.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def _lib():
    lib = ctypes.CDLL(build())
    lib.mt_state_size.restype = ctypes.c_size_t
    lib.mt_consumed.restype = ctypes.c_uint64
    return lib


class MT:
    """numpy-legacy-compatible MT19937 stream (state interoperable with RandomState)."""

    def __init__(self, seed=None):
        self.lib = _lib()
        self.buf = ctypes.create_string_buffer(self.lib.mt_state_size())
        if seed is not None:
            self.seed(seed)

    def seed(self, seed):
        self.lib.mt_seed(self.buf, ctypes.c_uint32(seed))

    def set_state(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        self.lib.mt_set_state(self.buf, key.ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(int(pos)))

    def get_state(self):
        key = np.empty(624, np.uint32)
        pos = ctypes.c_int32()
        self.lib.mt_get_state(self.buf, key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos))
        return key, pos.value

    def consumed(self):
        return int(self.lib.mt_consumed(self.buf))

    def randint(self, low, high, size):
        out = np.empty(size, np.int64)
        self.lib.mt_randint_fill(self.buf, ctypes.c_int64(low), ctypes.c_uint64(high - low - 1),
                                 ctypes.c_int64(size), out.ctypes.data_as(ctypes.c_void_p))
        return out

    def random_sample(self, size):
        out = np.empty(size, np.float64)
        self.lib.mt_random_sample_fill(self.buf, ctypes.c_int64(size), out.ctypes.data_as(ctypes.c_void_p))
        return out

    def her_draw(self, n_eps, T, B, future_p):
        e = np.empty(B, np.int64); t = np.empty(B, np.int64); fut = np.empty(B, np.int64)
        her = np.empty(B, np.uint8); u1 = np.empty(B, np.float64); u2 = np.empty(B, np.float64)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.lib.mt_her_draw(self.buf, ctypes.c_int64(n_eps), ctypes.c_int64(T), ctypes.c_int64(B),
                             ctypes.c_double(future_p), p(e), p(t), p(her), p(fut), p(u1), p(u2))
        return e, t, her.astype(bool), fut, u1, u2
