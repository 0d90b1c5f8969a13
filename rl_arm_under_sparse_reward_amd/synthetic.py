"""Seeded synthetic experience in the reference's episode layout (host-side, numpy only).

There is no PyBullet on the benchmark box, so bench.py, the tests and the golden
generator all draw episodes from here.  The recipe is the one BASELINE.md section 3
/ SURVEY.md section 8(d) fix for the headline metric:

    rs = RandomState(seed);  obs ~ U(-1,1) [N, T+1, 27];  ag = obs[:, :, 12:15]
    g  = one U(0,0.5)^3 goal per episode repeated over T;  actions ~ U(-0.5,0.5) [N, T, 4]

Shapes follow what ddpg_agent.py:138-143 hands to replay_buffer.store_episode:
obs [N,T+1,obs], ag [N,T+1,goal], g [N,T,goal], actions [N,T,action], all float64.

`mode="walk"` replaces the iid achieved goals by a slow random walk so that a useful
fraction of relabelled goals lands within the 0.05 success radius (exercises both
reward values and near-threshold distances in the parity tests).
"""
from __future__ import annotations

import numpy as np

ENV_PARAMS = {"obs": 27, "goal": 3, "action": 4, "action_max": 0.5, "max_timesteps": 100}


def make_episodes(n_episodes, seed=1, T=100, obs_dim=27, goal_dim=3, act_dim=4, mode="iid"):
    rs = np.random.RandomState(seed)
    obs = rs.uniform(-1.0, 1.0, size=(n_episodes, T + 1, obs_dim))
    if mode == "walk":
        start = rs.uniform(0.0, 0.5, size=(n_episodes, 1, goal_dim))
        steps = rs.normal(0.0, 0.012, size=(n_episodes, T + 1, goal_dim))
        steps[:, 0, :] = 0.0
        obs[:, :, 12:12 + goal_dim] = start + np.cumsum(steps, axis=1)
    elif mode != "iid":
        raise ValueError("mode must be 'iid' or 'walk'")
    ag = obs[:, :, 12:12 + goal_dim].copy()
    goal = rs.uniform(0.0, 0.5, size=(n_episodes, 1, goal_dim))
    g = np.repeat(goal, T, axis=1)
    actions = rs.uniform(-0.5, 0.5, size=(n_episodes, T, act_dim))
    return [np.ascontiguousarray(obs), ag, np.ascontiguousarray(g), np.ascontiguousarray(actions)]


def episode_checksum(episode_batch) -> float:
    """Exact, platform-independent checksum of the float64 bit patterns (integer arithmetic only, so it
    cannot depend on BLAS/libm code paths); returned as a float64-representable integer < 2^53."""
    acc = 0
    for i, a in enumerate(episode_batch):
        u = np.ascontiguousarray(a, dtype=np.float64).ravel().view(np.uint64)
        w = (np.arange(u.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(i + 1)) | np.uint64(1)
        with np.errstate(over="ignore"):
            acc = (acc + int(np.sum(u * w, dtype=np.uint64))) & ((1 << 64) - 1)
    return float(acc >> 11)


def write_demo_npz(path, n_episodes=8, seed=7, T=100):
    """Write a demo file in the schema get_demo_data_push.py:91-94 produces
    (keys acs, obs, info, g, ag; `info` is an object array of per-step dicts)."""
    obs, ag, g, actions = make_episodes(n_episodes, seed=seed, T=T, mode="walk")
    info = np.empty((n_episodes, T), dtype=object)
    for e in range(n_episodes):
        for t in range(T):
            d = float(np.linalg.norm(ag[e, t + 1] - g[e, t]))
            info[e, t] = {"is_success": np.float32(d < 0.05)}
    np.savez_compressed(path, acs=actions, obs=obs, info=info, g=g, ag=ag)
    return obs, ag, g, actions
