"""her_sampler -- drop-in mirror of the reference's her.py on the MI355X.

Same constructor and method as her.py:3-41:

    her_sampler(replay_strategy, replay_k, reward_func=None)
    .sample_her_transitions(episode_batch, batch_size_in_transitions) -> dict of ndarrays

but the index draw (her.py:24-33), gather (:26), goal relabel (:35-36) and reward (:38 ->
bmirobot_env_push_F.py:84-90) run as HIP kernels; see csrc/rng.hip and csrc/buffer.hip.

`reward_func` cannot be an arbitrary Python callable on the device.  The reference always
passes `env.compute_reward` of a bmirobot env whose reward is sparse with
`distance_threshold = 0.05` (bmirobot_push_F.py:9,20).  If the callable is a bound method
of an object with `distance_threshold` / `reward_type` attributes they are honoured;
anything else must be described with the keyword arguments.  Both branches of the envs'
compute_reward run on the device: 'sparse' (-(d > threshold)) and 'dense' (-d); any other
reward is refused loudly rather than silently computed on the host.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from . import random as _random

_KEYS = ("obs", "ag", "g", "actions", "obs_next", "ag_next")


def squared_threshold(distance_threshold: float) -> float:
    """Smallest float64 s with sqrt(s) > distance_threshold.

    `-(norm(a-b) > thr)` is evaluated on the device as `-(sum_sq >= s)`: sqrt is monotone and
    correctly rounded, so the two predicates agree for every input (checked against the
    reference on adversarial near-threshold pairs, tests/golden/reward_adversarial.npz)."""
    thr = float(distance_threshold)
    s = thr * thr
    while math.sqrt(s) > thr:
        s = math.nextafter(s, -math.inf)
    while not (math.sqrt(s) > thr):
        s = math.nextafter(s, math.inf)
    return s


class her_sampler:
    def __init__(self, replay_strategy, replay_k, reward_func=None, distance_threshold=None, reward_type=None,
                 rng=None):
        self.replay_strategy = replay_strategy
        self.replay_k = replay_k
        if self.replay_strategy == 'future':                       # her.py:7-10
            self.future_p = 1 - (1. / (1 + replay_k))
        else:
            self.future_p = 0
        self.reward_func = reward_func
        owner = getattr(reward_func, "__self__", None)
        if distance_threshold is None:
            distance_threshold = getattr(owner, "distance_threshold", 0.05)
        if reward_type is None:
            reward_type = getattr(owner, "reward_type", "sparse")
        if reward_type not in ("sparse", "dense"):
            raise NotImplementedError(
                "only the goal-distance rewards of the bmirobot tasks ('sparse', 'dense') run on the device "
                f"(got reward_type={reward_type!r}); there is no host fallback")
        self.reward_type = reward_type
        self.distance_threshold = float(distance_threshold)
        # the kernels take the squared threshold; a negative value selects the dense reward -d (compute_reward :89-90)
        self.sq_threshold = squared_threshold(self.distance_threshold) if reward_type == "sparse" else -1.0
        self._rng = rng

    @property
    def rng(self):
        return self._rng or _random.global_state()

    def sample_her_transitions(self, episode_batch, batch_size_in_transitions):
        """her.py:13-41 for a host episode dict (e.g. the two fresh episodes of
        ddpg_agent._update_normalizer).  The batch is staged in a scratch device buffer."""
        from .replay_buffer import DeviceEpisodeBuffer

        acts = np.asarray(episode_batch['actions'])
        n, T = acts.shape[0], acts.shape[1]
        obs = np.asarray(episode_batch['obs'])
        if obs.shape[1] != T + 1:
            raise ValueError("episode_batch['obs'] must have T+1 steps")
        if n == 0:
            raise ValueError("high <= 0")
        dev = DeviceEpisodeBuffer(n, T, obs.shape[2], np.asarray(episode_batch['g']).shape[2], acts.shape[2],
                                  ctx=self.rng.ctx)
        dev.store(self.rng, [obs, episode_batch['ag'], episode_batch['g'], acts])   # fits: draws nothing
        return dev.sample(self.rng, int(batch_size_in_transitions), self.future_p, self.sq_threshold)
