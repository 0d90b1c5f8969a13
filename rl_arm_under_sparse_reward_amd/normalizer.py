"""normalizer -- drop-in mirror of the reference's normalizer.py with state in HBM.

    normalizer(size, eps=1e-2, default_clip_range=np.inf)
    .update(v) .recompute_stats() .normalize(v, clip_range=None)   attrs: mean, std, ...

(normalizer.py:5-70).  The cross-rank mean of normalizer.py:34-38,60-64 (mpi4py Allreduce /
size) becomes an RCCL all-reduce over the library's device vector when torch.distributed is
initialised; single-process use needs no communicator.

`std` dtype: the reference's expression yields float64 under numpy >= 2 and float32 under the
numpy 1.19.2 it pins (README.md:11); by default this class reproduces what the reference
source would do under the numpy that is running (see oracle/running_norm.py for the probe).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _numpy_std_is_f32():
    return np.sqrt(np.maximum(np.square(1e-2), np.ones(1, np.float32))).dtype == np.float32


class normalizer:
    def __init__(self, size, eps=1e-2, default_clip_range=np.inf, std_dtype=None, ctx=None, comm=None):
        self.size = size
        self.eps = eps
        self.default_clip_range = default_clip_range
        self.ctx = ctx or _lib.Context.default()
        self.lib = self.ctx.lib
        if std_dtype is None:
            self._std_f32 = _numpy_std_is_f32()
        else:
            self._std_f32 = np.dtype(std_dtype) == np.float32
        self.comm = comm                     # utils.Communicator or None (single rank)
        self.h = C.c_void_p()
        clip = -1.0 if not np.isfinite(default_clip_range) else float(default_clip_range)
        _lib.check(self.lib.hp_norm_create(self.ctx.h, int(size), float(eps), float("inf") if clip < 0 else clip,
                                           int(self._std_f32), C.byref(self.h)))

    # ---- reference API
    def update(self, v):
        v = _lib.as_f64(v).reshape(-1, self.size)                       # normalizer.py:26
        _lib.check(self.lib.hp_norm_update(self.h, _lib.ptr(v, C.c_double), v.shape[0]))

    def recompute_stats(self):
        if self.comm is None or not self.comm.active:
            _lib.check(self.lib.hp_norm_recompute(self.h))
            return
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.hp_norm_recompute_begin(self.h, C.byref(p), C.byref(n)))
        self.comm.allreduce_mean_device(p.value, n.value)               # normalizer.py:60-64
        _lib.check(self.lib.hp_norm_recompute_end(self.h))

    def sync(self, local_sum, local_sumsq, local_count):
        """normalizer.py:34-38: the cross-rank average of three host arrays (recompute_stats does this on the device
        vectors; kept for callers of the reference's helper)."""
        local_sum[...] = self._mpi_average(local_sum)
        local_sumsq[...] = self._mpi_average(local_sumsq)
        local_count[...] = self._mpi_average(local_count)
        return local_sum, local_sumsq, local_count

    def _mpi_average(self, x):
        """normalizer.py:60-64: Allreduce(SUM) / world size."""
        import torch
        buf = torch.from_numpy(np.array(x, dtype=np.float32, copy=True))
        if self.comm is not None and self.comm.active:
            self.comm.allreduce_mean_(buf)
        return buf.numpy()

    def normalize(self, v, clip_range=None):
        if clip_range is None:
            clip_range = self.default_clip_range
        a = _lib.as_f64(v)
        flat = a.reshape(-1, self.size)
        out = np.empty_like(flat)
        clip = float(clip_range)
        _lib.check(self.lib.hp_norm_normalize(self.h, _lib.ptr(flat, C.c_double), flat.shape[0], clip,
                                              _lib.ptr(out, C.c_double)))
        return out.reshape(a.shape)

    # ---- state (host copies; the reference exposes these as numpy attributes and checkpoints
    #      mean/std at ddpg_agent.py:158-161)
    def _get(self):
        n = self.size
        d = {k: np.empty(n, np.float32) for k in ("mean", "total_sum", "total_sumsq", "local_sum", "local_sumsq")}
        d["std"] = np.empty(n, np.float64)
        d["total_count"], d["local_count"] = np.empty(1, np.float32), np.empty(1, np.float32)
        f = C.c_float
        _lib.check(self.lib.hp_norm_get(self.h, _lib.ptr(d["mean"], f), _lib.ptr(d["std"], C.c_double),
                                        _lib.ptr(d["total_sum"], f), _lib.ptr(d["total_sumsq"], f),
                                        _lib.ptr(d["total_count"], f), _lib.ptr(d["local_sum"], f),
                                        _lib.ptr(d["local_sumsq"], f), _lib.ptr(d["local_count"], f)))
        if self._std_f32:
            d["std"] = d["std"].astype(np.float32)      # exact: the device value is a widened float32
        return d

    mean = property(lambda self: self._get()["mean"])
    std = property(lambda self: self._get()["std"])
    total_sum = property(lambda self: self._get()["total_sum"])
    total_sumsq = property(lambda self: self._get()["total_sumsq"])
    total_count = property(lambda self: self._get()["total_count"])
    local_sum = property(lambda self: self._get()["local_sum"])
    local_sumsq = property(lambda self: self._get()["local_sumsq"])
    local_count = property(lambda self: self._get()["local_count"])

    def set_stats(self, mean, std):
        """Load checkpointed statistics (demo_push.py:28,41 reads them back the same way)."""
        m = _lib.as_f32(mean)
        s = _lib.as_f64(std)
        _lib.check(self.lib.hp_norm_set_stats(self.h, _lib.ptr(m, C.c_float), _lib.ptr(s, C.c_double)))

    def __del__(self):
        try:
            self.lib.hp_norm_destroy(self.h)
        except Exception:
            pass
