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


class PointMassGoalEnv:
    """Stand-in for the PyBullet arm environments (the real ones need pybullet + gym, absent here): the
    gym GoalEnv surface the learner's rollout loop uses -- reset() / step(action) returning the dict observation
    {'observation', 'achieved_goal', 'desired_goal'} (bmirobot_env_push_F.py:233-237), step's 4-tuple with
    info['is_success'] (:103-108) and a compute_reward vectorised over leading dims (:84-90) -- around a point mass
    that the first three action components push through a 0.5 m box.  The observation keeps the bmirobot layout
    (27 = 9 blocks of 3, achieved goal = block 4, :214,228).  It has its own RandomState so it never draws from
    numpy's global stream, which belongs to the exploration noise."""

    def __init__(self, seed=0, max_timesteps=100, distance_threshold=0.05, reward_type='sparse', step_scale=0.1):
        self.rs = np.random.RandomState(seed)
        self.max_timesteps = int(max_timesteps)
        self.distance_threshold = float(distance_threshold)
        self.reward_type = reward_type
        self.step_scale = float(step_scale)
        self.pos = np.zeros(3)
        self.vel = np.zeros(3)
        self.goal = np.zeros(3)

    @property
    def env_params(self):
        return {'obs': 27, 'goal': 3, 'action': 4, 'action_max': 0.5, 'max_timesteps': self.max_timesteps}

    def _observation(self):
        obs = np.zeros(27)
        obs[0:3] = self.pos                  # "gripper" block
        obs[3:6] = self.vel
        obs[12:15] = self.pos                # the block whose position is the achieved goal
        return {'observation': obs, 'achieved_goal': self.pos.copy(), 'desired_goal': self.goal.copy()}

    def reset(self):
        self.pos = self.rs.uniform(0.0, 0.5, 3)
        self.goal = self.rs.uniform(0.0, 0.5, 3)
        self.vel = np.zeros(3)
        return self._observation()

    def compute_reward(self, achieved_goal, goal, info):
        diff = np.asarray(achieved_goal) - np.asarray(goal)
        d = np.linalg.norm(diff, axis=-1)
        if self.reward_type == 'sparse':
            return -(d > self.distance_threshold).astype(np.float32)
        return -d

    def _is_success(self, achieved_goal, desired_goal):
        return (np.linalg.norm(achieved_goal - desired_goal, axis=-1) < self.distance_threshold).astype(np.float32)

    def step(self, action):
        action = np.clip(np.asarray(action, dtype=np.float64), -0.5, 0.5)
        new = np.clip(self.pos + self.step_scale * action[:3], 0.0, 0.5)
        self.vel = new - self.pos
        self.pos = new
        observation = self._observation()
        info = {'is_success': self._is_success(observation['achieved_goal'], self.goal)}
        reward = self.compute_reward(observation['achieved_goal'], self.goal, info)
        return observation, reward, False, info
