"""Device-resident twin of numpy's legacy *global* RandomState.

The reference draws every sampled index from the process-global `np.random`
(her.py:24-31, replay_buffer.py:64,67), seeded with `seed + rank` (train.py:36).  Here
the stream lives in HBM next to the kernels that consume it; this module mirrors the
numpy calls a reference-style script uses to control it:

    seed(s)          <-> np.random.seed(s)
    get_state()      <-> np.random.get_state()      (same 5-tuple)
    set_state(st)    <-> np.random.set_state(st)

so a script can hand the stream back and forth, e.g. `random.set_state(np.random.get_state())`
before the learner phase and `np.random.set_state(random.get_state())` after it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class DeviceRandomState:
    def __init__(self, seed=None, ctx=None):
        self.ctx = ctx or _lib.Context.default()
        self.lib = self.ctx.lib
        self.h = C.c_void_p()
        _lib.check(self.lib.hp_rng_create(self.ctx.h, C.byref(self.h)))
        self.seeded = False       # seed() / set_state() called (a fresh stream carries numpy's default key, seed 5489)
        # numpy's legacy state also carries a cached Gaussian (has_gauss, cached_gaussian).  Nothing on the device draws
        # normals, and randint / random_sample never touch that cache in numpy either, so it is carried through unchanged:
        # np.random.set_state(dev.get_state()) after dev.set_state(np.random.get_state()) keeps a pending second normal
        self._gauss = (0, 0.0)
        if seed is not None:
            self.seed(seed)

    def seed(self, seed):
        seed = int(seed)
        if not 0 <= seed <= 2**32 - 1:
            raise ValueError("Seed must be between 0 and 2**32 - 1")     # numpy's message
        _lib.check(self.lib.hp_rng_seed(self.h, C.c_uint32(seed)))
        self.seeded = True
        self._gauss = (0, 0.0)                                           # numpy's seed() drops the cached normal

    def get_state(self):
        key = np.empty(624, np.uint32)
        pos = C.c_int32()
        _lib.check(self.lib.hp_rng_get_state(self.h, _lib.ptr(key, C.c_uint32), C.byref(pos)))
        return ("MT19937", key, int(pos.value), int(self._gauss[0]), float(self._gauss[1]))

    def set_state(self, state):
        if isinstance(state, dict):
            key, pos = state["state"]["key"], state["state"]["pos"]
            gauss = (int(state.get("has_gauss", 0)), float(state.get("gauss", 0.0)))
        else:
            if state[0] != "MT19937":
                raise ValueError("set_state can only be used with legacy MT19937 state instances.")
            key, pos = state[1], state[2]
            gauss = (int(state[3]), float(state[4])) if len(state) >= 5 else (0, 0.0)   # numpy accepts the 3-tuple too
        key = np.ascontiguousarray(key, dtype=np.uint32)
        if key.shape != (624,):
            raise ValueError("state must be 624 longs")
        _lib.check(self.lib.hp_rng_set_state(self.h, _lib.ptr(key, C.c_uint32), C.c_int32(int(pos))))
        self.seeded = True
        self._gauss = gauss

    # test hooks: the two primitive draws of the hot path, executed on the device
    def randint(self, low, high=None, size=1):
        if high is None:
            low, high = 0, low
        out = np.empty(int(size), np.int64)
        _lib.check(self.lib.hp_rng_randint(self.h, int(low), int(high), int(size), _lib.ptr(out, C.c_int64)))
        return out

    def uniform(self, size=1):
        out = np.empty(int(size), np.float64)
        _lib.check(self.lib.hp_rng_uniform(self.h, int(size), _lib.ptr(out, C.c_double)))
        return out

    def __del__(self):
        try:
            self.lib.hp_rng_destroy(self.h)
        except Exception:
            pass


_global = None


def global_state() -> DeviceRandomState:
    """The process-global stream (created on first use), like numpy's `np.random` singleton."""
    global _global
    if _global is None:
        _global = DeviceRandomState()
    return _global


def seed(s):
    global_state().seed(s)


def get_state():
    return global_state().get_state()


def set_state(state):
    global_state().set_state(state)
