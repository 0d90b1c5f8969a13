"""oracle/rollout.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy/torch-CPU restatement of the ROLLOUT half of the reference's ddpg_agent.py -- the step immediately upstream
(experience collection) and downstream (evaluation) of the hot path, SURVEY.md 8(f) N1 / N3:

  * _preproc_inputs                 ddpg_agent.py:163-171
  * _select_actions                 ddpg_agent.py:174-184   (draw order: randn(action), uniform(action), binomial(1))
  * the rollout loop of learn()     ddpg_agent.py:101-142   (+ the +-0.15 action clip from epoch 100, :118-119)
  * _eval_agent                     ddpg_agent.py:280-304   (success of the LAST step of each test rollout, rank mean)
  * learn() as a whole              ddpg_agent.py:92-161    (`learn_epochs`, on the oracle learner / store / normalizers)

Pinned by tests/golden/rollout.npz: tools/gen_golden.py runs the reference's own learn() and _eval_agent() on the
stand-in GoalEnv of the package (the PyBullet envs need gym + pybullet) and this module reproduces the stored episodes,
success rates, RNG state and parameters bit for bit (tests/test_oracle_rollout.py).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ddpg_update as oupd
from .her_replay import EpisodeStore, compute_reward, future_probability
from .running_norm import RunningNorm, update_normalizers


def preproc_inputs(o_norm, g_norm, obs, g):
    """ddpg_agent.py:163-171."""
    inputs = np.concatenate([o_norm.normalize(obs), g_norm.normalize(g)])
    return torch.tensor(inputs, dtype=torch.float32).unsqueeze(0)


def select_actions(pi, noise_eps, random_eps, action_max, n_action, rs=np.random):
    """ddpg_agent.py:174-184.  `pi` float32 tensor [1, action]; the float32 array is updated in place."""
    action = pi.cpu().numpy().squeeze()
    action += noise_eps * action_max * rs.randn(*action.shape)
    action = np.clip(action, -action_max, action_max)
    random_actions = rs.uniform(low=-action_max, high=action_max, size=n_action)
    action += rs.binomial(1, random_eps, 1)[0] * (random_actions - action)
    return action


def rollout_episode(env, policy, T, epoch=0, explore=None):
    """ddpg_agent.py:105-137 for one rollout.  policy(obs, g) -> float32 tensor [1, action];
    explore(pi) -> action (None: the raw policy output, as _eval_agent uses it)."""
    ep_obs, ep_ag, ep_g, ep_actions = [], [], [], []
    observation = env.reset()
    obs, ag, g = observation['observation'], observation['achieved_goal'], observation['desired_goal']
    for _ in range(int(T)):
        with torch.no_grad():
            pi = policy(obs, g)
            action = explore(pi) if explore is not None else pi.detach().cpu().numpy().squeeze()
        if epoch >= 100:
            action = np.clip(action, -0.15, 0.15)              # :118-119
        observation_new, _, _, info = env.step(action)
        ep_obs.append(obs.copy()); ep_ag.append(ag.copy()); ep_g.append(g.copy()); ep_actions.append(action.copy())
        obs, ag = observation_new['observation'], observation_new['achieved_goal']
    ep_obs.append(obs.copy()); ep_ag.append(ag.copy())
    return ep_obs, ep_ag, ep_g, ep_actions


def eval_agent(env, policy, n_test_rollouts, T, mean_over_ranks=None):
    """ddpg_agent.py:280-304."""
    total = []
    for _ in range(n_test_rollouts):
        per = []
        observation = env.reset()
        obs, g = observation['observation'], observation['desired_goal']
        for _ in range(T):
            with torch.no_grad():
                actions = policy(obs, g).detach().cpu().numpy().squeeze()
            observation_new, _, _, info = env.step(actions)
            obs, g = observation_new['observation'], observation_new['desired_goal']
            per.append(info['is_success'])
        total.append(per)
    local = np.mean(np.array(total)[:, -1])
    return mean_over_ranks(local) if mean_over_ranks else local


class OracleAgent:
    """The reference's ddpg_agent (both halves) on the oracle pieces; one rank."""

    def __init__(self, env, env_params, actor, critic, *, buffer_size, batch_size=256, replay_k=4, n_batches=40,
                 num_rollouts=2, n_test_rollouts=25, noise_eps=0.01, random_eps=0.3, clip_obs=200, clip_range=5,
                 rs=np.random):
        # her.py:38 calls env.compute_reward: the env's own threshold / reward type (bmirobot_push_F.py:9,20 -> 0.05, sparse)
        thr, kind = getattr(env, "distance_threshold", 0.05), getattr(env, "reward_type", "sparse")
        self.reward_fn = lambda a, g: compute_reward(a, g, thr, kind)
        self.env, self.p = env, env_params
        self.T = int(env_params['max_timesteps'])
        self.learner = oupd.DDPGLearner(actor, critic, max_action=env_params['action_max'])
        self.store = EpisodeStore(self.T, env_params['obs'], env_params['goal'], env_params['action'], buffer_size)
        self.fp = future_probability("future", replay_k)
        self.o_norm = RunningNorm(env_params['obs'], default_clip_range=clip_range)
        self.g_norm = RunningNorm(env_params['goal'], default_clip_range=clip_range)
        self.batch_size, self.n_batches, self.num_rollouts, self.n_test_rollouts = batch_size, n_batches, num_rollouts, n_test_rollouts
        self.noise_eps, self.random_eps, self.clip_obs, self.rs = noise_eps, random_eps, clip_obs, rs
        self.episodes, self.success_rates = [], []

    def policy(self, obs, g):
        return oupd.actor_forward(self.learner.actor, preproc_inputs(self.o_norm, self.g_norm, obs, g),
                                  self.p['action_max'])

    def _explore(self, pi):
        return select_actions(pi, self.noise_eps, self.random_eps, self.p['action_max'], self.p['action'], self.rs)

    def cycle(self, epoch=0):
        """ddpg_agent.py:101-150."""
        mb = ([], [], [], [])
        for _ in range(self.num_rollouts):
            for dst, src in zip(mb, rollout_episode(self.env, self.policy, self.T, epoch, self._explore)):
                dst.append(src)
        batch = [np.array(a) for a in mb]
        self.episodes.append(batch)
        self.store.store_episode(batch, self.rs)
        update_normalizers(self.o_norm, self.g_norm, batch, self.fp, self.rs, self.clip_obs)
        for _ in range(self.n_batches):
            tr, _ = self.store.sample(self.batch_size, self.fp, self.rs, self.reward_fn)
            self.learner.update(*oupd.minibatch_tensors(tr, self.o_norm, self.g_norm, self.clip_obs))
        self.learner.soft_update()

    def learn_epochs(self, n_epochs, n_cycles):
        for epoch in range(n_epochs):
            for _ in range(n_cycles):
                self.cycle(epoch)
            self.success_rates.append(eval_agent(self.env, self.policy, self.n_test_rollouts, self.T))
