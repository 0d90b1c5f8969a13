"""oracle/running_norm.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of normalizer.py:5-70 and of ddpg_agent.py:187-217 (the
`_update_normalizer` / `_preproc_og` callers).

dtype note (pinned by tests/golden/normalizer.npz, generated from the reference on this
container's numpy 2.2.6): accumulators, totals and `mean` are float32.  `std` is
  sqrt(maximum(np.square(eps), total_sumsq/total_count - square(total_sum/total_count)))
where np.square(eps) is a *numpy float64 scalar*.  Under numpy >= 2 (NEP 50) that scalar
is strongly typed, so maximum() and sqrt() run in float64 and `std` is a float64 array;
under the numpy 1.19.2 the reference pins (README.md:11) value-based casting keeps the
whole expression float32.  `std_dtype` selects which of the two is restated; the default
follows the numpy that is running, exactly like the reference source would.
"""
from __future__ import annotations

import numpy as np


def numpy_std_dtype():
    """dtype the reference's `std` expression yields under the running numpy."""
    return np.sqrt(np.maximum(np.square(1e-2), np.ones(1, np.float32))).dtype


class RunningNorm:
    """normalizer.py:5-70 with an explicit cross-rank mean hook.

    `allreduce_mean(x)` must return the element-wise mean of x over ranks
    (normalizer.py:60-64: Allreduce(SUM) then divide by Get_size()); identity by default.
    """

    def __init__(self, size, eps=1e-2, default_clip_range=np.inf, allreduce_mean=None, std_dtype=None):
        self.size = size
        self.eps = eps
        self.default_clip_range = default_clip_range
        self.local_sum = np.zeros(size, np.float32)
        self.local_sumsq = np.zeros(size, np.float32)
        self.local_count = np.zeros(1, np.float32)
        self.total_sum = np.zeros(size, np.float32)
        self.total_sumsq = np.zeros(size, np.float32)
        self.total_count = np.ones(1, np.float32)            # normalizer.py:17 -- starts at ONE
        self.mean = np.zeros(size, np.float32)
        self.std = np.ones(size, np.float32)
        self._mean_over_ranks = allreduce_mean or (lambda x: x.copy())
        self._std_dtype = np.dtype(std_dtype) if std_dtype is not None else numpy_std_dtype()

    def update(self, v):
        """normalizer.py:25-31: float64 column sums added into float32 accumulators."""
        v = v.reshape(-1, self.size)
        self.local_sum += v.sum(axis=0)
        self.local_sumsq += np.square(v).sum(axis=0)
        self.local_count[0] += v.shape[0]

    def recompute_stats(self):
        """normalizer.py:40-57."""
        c, s, ss = self.local_count.copy(), self.local_sum.copy(), self.local_sumsq.copy()
        self.local_count[...] = 0
        self.local_sum[...] = 0
        self.local_sumsq[...] = 0
        s = self._mean_over_ranks(s)            # normalizer.py:35-37 (order: sum, sumsq, count)
        ss = self._mean_over_ranks(ss)
        c = self._mean_over_ranks(c)
        self.total_sum += s
        self.total_sumsq += ss
        self.total_count += c
        self.mean = self.total_sum / self.total_count
        var = (self.total_sumsq / self.total_count) - np.square(self.total_sum / self.total_count)
        if self._std_dtype == np.float64:
            self.std = np.sqrt(np.maximum(np.float64(np.square(self.eps)), var.astype(np.float64)))
        else:
            self.std = np.sqrt(np.maximum(np.float32(np.square(self.eps)), var))

    def normalize(self, v, clip_range=None):
        """normalizer.py:67-70."""
        if clip_range is None:
            clip_range = self.default_clip_range
        return np.clip((v - self.mean) / self.std, -clip_range, clip_range)


def preproc_og(o, g, clip_obs=200):
    """ddpg_agent.py:214-217."""
    return np.clip(o, -clip_obs, clip_obs), np.clip(g, -clip_obs, clip_obs)


def update_normalizers(o_norm, g_norm, episode_batch, future_p, rng, clip_obs=200):
    """ddpg_agent.py:187-212: HER-sample T transitions from the fresh episodes, clip,
    feed both normalizers, recompute.  Returns the sampled (obs, g) for inspection."""
    from .her_replay import sample_her_transitions

    mb_obs, mb_ag, mb_g, mb_actions = episode_batch
    tmp = {
        "obs": mb_obs, "ag": mb_ag, "g": mb_g, "actions": mb_actions,
        "obs_next": mb_obs[:, 1:, :], "ag_next": mb_ag[:, 1:, :],
    }
    num_transitions = mb_actions.shape[1]                      # ddpg_agent.py:194 -- always T
    tr, _ = sample_her_transitions(tmp, num_transitions, future_p, rng)
    o, g = preproc_og(tr["obs"], tr["g"], clip_obs)
    o_norm.update(o)
    g_norm.update(g)
    o_norm.recompute_stats()
    g_norm.recompute_stats()
    return o, g
