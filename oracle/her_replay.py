"""oracle/her_replay.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of the reference's data path:
  * sparse goal-distance reward   bmirobot_env/bmirobot_env_push_F.py:20-23,84-90
                                   (byte-identical in bmirobot_env_pickandplace_v2.py:20-23,84-90)
  * HER 'future' relabelling      her.py:4-41
  * episodic replay storage       replay_buffer.py:11-71

Differences from the reference are structural only: the RNG is an explicit
np.random.RandomState instead of the process-global one, and the index draw, gather,
relabel and reward steps are separate functions so each can be compared with one
device kernel stage.  Arithmetic, dtypes and RNG consumption order are the same.
"""
from __future__ import annotations

import numpy as np

EPISODE_KEYS = ("obs", "ag", "g", "actions")


# --------------------------------------------------------------------------- reward
def goal_distance(goal_a, goal_b):
    """bmirobot_env_push_F.py:20-23 -- L2 norm over the last axis, float64."""
    if goal_a.shape != goal_b.shape:
        raise AssertionError("goal shapes differ")
    return np.linalg.norm(goal_a - goal_b, axis=-1)


def compute_reward(achieved_goal, goal, distance_threshold=0.05, reward_type="sparse"):
    """bmirobot_env_push_F.py:84-90.  sparse: -(d > thr) as float32 (-0.0 on success)."""
    d = goal_distance(achieved_goal, goal)
    if reward_type == "sparse":
        return -(d > distance_threshold).astype(np.float32)
    return -d


def is_success(achieved_goal, desired_goal, distance_threshold=0.05):
    """bmirobot_env_push_F.py:243-245 -- (d < thr) as float32."""
    d = goal_distance(achieved_goal, desired_goal)
    return (d < distance_threshold).astype(np.float32)


def squared_distance_threshold(distance_threshold: float) -> float:
    """Smallest float64 s with sqrt(s) > distance_threshold (sqrt correctly rounded).

    sqrt_rn is monotone, so `sqrt(s) > thr` <=> `s >= s*`.  The device kernel compares the
    squared distance with s* and never takes a square root (hard
    parts).  math.sqrt is correctly rounded on IEEE hosts.
    """
    import math

    thr = float(distance_threshold)
    s = thr * thr
    # walk down while still above, then up until strictly above
    while math.sqrt(s) > thr:
        s = math.nextafter(s, -math.inf)
    while not (math.sqrt(s) > thr):
        s = math.nextafter(s, math.inf)
    return s


# ------------------------------------------------------------------------ HER sampler
def future_probability(replay_strategy: str, replay_k: int) -> float:
    """her.py:7-10."""
    if replay_strategy == "future":
        return 1 - (1.0 / (1 + replay_k))
    return 0


def draw_her_indices(rng: np.random.RandomState, n_episodes: int, T: int, batch: int, future_p: float):
    """her.py:24,25,29,31-33 -- the four draws, in stream order.

    Returns (episode_idxs, t_samples, her_mask, future_t_all) where future_t_all is
    defined for every sample (the reference only keeps it where her_mask is set).
    """
    episode_idxs = rng.randint(0, n_episodes, batch)
    t_samples = rng.randint(T, size=batch)
    her_mask = rng.uniform(size=batch) < future_p
    future_offset = (rng.uniform(size=batch) * (T - t_samples)).astype(int)
    future_t_all = t_samples + 1 + future_offset
    return episode_idxs, t_samples, her_mask, future_t_all


# ---- fast draw (the build's opt-in rng_mode of SURVEY 8b; NOT the reference's stream): counter-based indices ----------------------
PHILOX_M0, PHILOX_M1, PHILOX_W0, PHILOX_W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 of Random123 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC11), vectorised
    over numpy arrays of uint64 holding 32-bit values.  Pinned by the package's known-answer vectors (tests/test_oracle_rng.py):
    counter 0, key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8; all ones -> 408f276d 41c83b0e a20bc7c6 6d5451fd; digits of pi ->
    d16cfe09 94fdcceb 5001e420 24126ea1."""
    u, mask = np.uint64, np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3))
    k0, k1 = u(k0), u(k1)
    for _ in range(10):
        p0, p1 = u(PHILOX_M0) * c0, u(PHILOX_M1) * c2
        c0, c1, c2, c3 = (p1 >> u(32)) ^ c1 ^ k0, p1 & mask, (p0 >> u(32)) ^ c3 ^ k1, p0 & mask
        k0, k1 = (k0 + u(PHILOX_W0)) & mask, (k1 + u(PHILOX_W1)) & mask
    return c0, c1, c2, c3


def draw_her_indices_fast(n_episodes: int, T: int, batch: int, future_p: float, seed: int, call: int):
    """The four draws of her.py:24-33 for transition m of call `call` from Philox4x32-10(counter (m, call), key seed):
    e = floor(r0 N / 2^32), t = floor(r1 T / 2^32), her = r2 2^-32 < future_p, future_t = t + 1 + floor(r3 (T - t) / 2^32)
    (csrc/buffer.hip: fs_fast_rec is the device twin)."""
    m = np.arange(batch, dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    r0, r1, r2, r3 = philox4x32_10(m & mask, m >> np.uint64(32), np.uint64(call & 0xFFFFFFFF), np.uint64((call >> 32) & 0xFFFFFFFF),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    e = ((r0 * np.uint64(n_episodes)) >> np.uint64(32)).astype(np.int64)
    t = ((r1 * np.uint64(T)) >> np.uint64(32)).astype(np.int64)
    her = (r2.astype(np.float64) * 2.0 ** -32) < future_p
    future_t = t + 1 + ((r3 * (np.uint64(T) - t.astype(np.uint64))) >> np.uint64(32)).astype(np.int64)
    return e, t, her, future_t


def sample_her_transitions(episode_batch, batch, future_p, rng, reward_fn=compute_reward, indices=None):
    """her.py:13-41 on a dict holding obs/ag/g/actions/obs_next/ag_next.  indices: (e, t, her_mask, future_t) drawn elsewhere
    (the fast draw) instead of from `rng`."""
    T = episode_batch["actions"].shape[1]
    n_episodes = episode_batch["actions"].shape[0]
    e, t, her_mask, future_t = indices if indices is not None else draw_her_indices(rng, n_episodes, T, batch, future_p)
    out = {k: v[e, t].copy() for k, v in episode_batch.items()}          # her.py:26
    sel = np.where(her_mask)
    out["g"][sel] = episode_batch["ag"][e[sel], future_t[sel]]           # her.py:35-36
    out["r"] = np.expand_dims(reward_fn(out["ag_next"], out["g"]), 1)    # her.py:38
    out = {k: v.reshape(batch, *v.shape[1:]) for k, v in out.items()}    # her.py:39
    return out, dict(e=e, t=t, her=her_mask, future_t=future_t)


def with_next_views(buffers, current_size):
    """replay_buffer.py:48-52 -- slice to the filled part and add the t+1 views."""
    tmp = {k: buffers[k][:current_size] for k in EPISODE_KEYS}
    tmp["obs_next"] = tmp["obs"][:, 1:, :]
    tmp["ag_next"] = tmp["ag"][:, 1:, :]
    return tmp


# --------------------------------------------------------------------- episodic buffer
class EpisodeStore:
    """replay_buffer.py:11-71 (storage policy + sampling wrapper), explicit RNG."""

    def __init__(self, T, obs_dim, goal_dim, act_dim, buffer_size):
        self.T = int(T)
        self.size = int(buffer_size // self.T)                     # replay_buffer.py:16
        self.current_size = 0
        self.n_transitions_stored = 0
        self.buffers = {
            "obs": np.empty([self.size, self.T + 1, obs_dim]),
            "ag": np.empty([self.size, self.T + 1, goal_dim]),
            "g": np.empty([self.size, self.T, goal_dim]),
            "actions": np.empty([self.size, self.T, act_dim]),
        }

    def storage_slots(self, inc, rng):
        """replay_buffer.py:57-71.  Returns an int64 array (the reference returns a scalar
        when inc == 1; callers index with it either way)."""
        inc = inc or 1
        cur, size = self.current_size, self.size
        if cur + inc <= size:
            idx = np.arange(cur, cur + inc)
        elif cur < size:
            overflow = inc - (size - cur)
            idx = np.concatenate([np.arange(cur, size), rng.randint(0, cur, overflow)])
        else:
            idx = rng.randint(0, size, inc)
        self.current_size = min(size, cur + inc)
        return idx

    def store_episode(self, episode_batch, rng):
        """replay_buffer.py:32-43."""
        mb_obs, mb_ag, mb_g, mb_actions = episode_batch
        n = mb_obs.shape[0]
        idx = self.storage_slots(n, rng)
        self.buffers["obs"][idx] = mb_obs
        self.buffers["ag"][idx] = mb_ag
        self.buffers["g"][idx] = mb_g
        self.buffers["actions"][idx] = mb_actions
        self.n_transitions_stored += self.T * n
        return idx

    def sample(self, batch, future_p, rng, reward_fn=compute_reward):
        """replay_buffer.py:46-55."""
        return sample_her_transitions(with_next_views(self.buffers, self.current_size), batch, future_p, rng, reward_fn)

    def sample_fast(self, batch, future_p, seed, call, reward_fn=compute_reward):
        """The same gather / relabel / reward on the fast draw's indices (draw_her_indices_fast)."""
        idx = draw_her_indices_fast(self.current_size, self.T, batch, future_p, seed, call)
        return sample_her_transitions(with_next_views(self.buffers, self.current_size), batch, future_p, None, reward_fn, indices=idx)
