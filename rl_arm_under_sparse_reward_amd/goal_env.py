"""compute_reward / _is_success of the bmirobot GoalEnvs as batched device ops.

The reference envs define (bmirobot_env/bmirobot_env_push_F.py:20-23,84-90,243-245; byte-identical in
bmirobot_env_pickandplace_v2.py):

    goal_distance(a, b)              = np.linalg.norm(a - b, axis=-1)
    compute_reward(ag, g, info)      = -(d > distance_threshold).astype(np.float32)   'sparse'
                                     = -d                                             otherwise ('dense')
    _is_success(ag, g)               = (d < distance_threshold).astype(np.float32)

`GoalDistanceReward` keeps those call signatures (vectorised over leading dims, as her.py:38 relies on) and runs
them through `hp_compute_reward` / `hp_is_success` (csrc/buffer.hip:k_goal_reward): float64 squared distance summed
left to right, predicates evaluated against the exact squared-domain bound, so the results carry the same bits as the
numpy expressions.  A GoalEnv wrapper can delegate to it (`env.compute_reward = reward.compute_reward`), and
`her_sampler` reads `distance_threshold` / `reward_type` from it like from a reference env.  There is no host path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class GoalDistanceReward:
    def __init__(self, distance_threshold=0.05, reward_type="sparse", ctx=None):
        if reward_type not in ("sparse", "dense"):
            raise NotImplementedError(f"reward_type={reward_type!r}: only 'sparse' and 'dense' (compute_reward :84-90)")
        self.distance_threshold = float(distance_threshold)
        self.reward_type = reward_type
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = _lib.Context.default()
        return self._ctx

    @staticmethod
    def _pairs(achieved_goal, goal):
        a, g = _lib.as_f64(achieved_goal), _lib.as_f64(goal)
        assert a.shape == g.shape                                  # goal_distance :21
        if a.ndim == 0:
            raise ValueError("goal arrays need at least one axis")
        return a, g, a.shape[:-1], int(a.shape[-1])

    def compute_reward(self, achieved_goal, goal, info=None):
        """bmirobot_env_push_F.py:84-90: float32 -(d > thr) ('sparse') or float64 -d ('dense'), shape = leading dims."""
        a, g, lead, gd = self._pairs(achieved_goal, goal)
        n = int(np.prod(lead, dtype=np.int64)) if lead else 1
        dense = self.reward_type != "sparse"
        out = np.empty(n, np.float64 if dense else np.float32)
        _lib.check(self.ctx.lib.hp_compute_reward(
            self.ctx.h, _lib.ptr(a, C.c_double), _lib.ptr(g, C.c_double), n, gd, self.distance_threshold, int(dense),
            None if dense else _lib.ptr(out, C.c_float), _lib.ptr(out, C.c_double) if dense else None))
        return out.reshape(lead) if lead else out.reshape(())[()]

    def _is_success(self, achieved_goal, desired_goal):
        """bmirobot_env_push_F.py:243-245: float32 (d < thr)."""
        a, g, lead, gd = self._pairs(achieved_goal, desired_goal)
        n = int(np.prod(lead, dtype=np.int64)) if lead else 1
        out = np.empty(n, np.float32)
        _lib.check(self.ctx.lib.hp_is_success(self.ctx.h, _lib.ptr(a, C.c_double), _lib.ptr(g, C.c_double), n, gd,
                                              self.distance_threshold, _lib.ptr(out, C.c_float)))
        return out.reshape(lead) if lead else out.reshape(())[()]

    is_success = _is_success

    # device-array forms (torch tensors or raw addresses on this context's device); asynchronous on the context's stream
    def compute_reward_device(self, ag_ptr, g_ptr, n, goal_dim, out_ptr):
        dense = self.reward_type != "sparse"
        _lib.check(self.ctx.lib.hp_compute_reward_dev(
            self.ctx.h, C.c_void_p(int(ag_ptr)), C.c_void_p(int(g_ptr)), int(n), int(goal_dim), self.distance_threshold,
            int(dense), None if dense else C.c_void_p(int(out_ptr)), C.c_void_p(int(out_ptr)) if dense else None))

    def is_success_device(self, ag_ptr, g_ptr, n, goal_dim, out_ptr):
        _lib.check(self.ctx.lib.hp_is_success_dev(self.ctx.h, C.c_void_p(int(ag_ptr)), C.c_void_p(int(g_ptr)), int(n),
                                                  int(goal_dim), self.distance_threshold, C.c_void_p(int(out_ptr))))
