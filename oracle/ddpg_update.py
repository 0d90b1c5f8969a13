"""oracle/ddpg_update.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch-CPU restatement of the reference learner:
  * actor / critic MLPs                   models.py:11-44
  * one DDPG update                        ddpg_agent.py:225-277
  * polyak target update                   ddpg_agent.py:220-222
  * flat parameter / gradient exchange     utils.py:6-69  (Bcast root 0 / Allreduce SUM)

The networks are plain dicts of tensors keyed like the reference's state_dict
(`fc1.weight` ... `action_out.bias` / `q_out.bias`) and applied functionally;
torch.optim.Adam, autograd and F.linear/relu/tanh are the same third-party arithmetic
the reference calls.  This module is also the CPU baseline ("port") bench.py times.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

HIDDEN = 256
ACTOR_KEYS = ("fc1", "fc2", "fc3", "action_out")
CRITIC_KEYS = ("fc1", "fc2", "fc3", "q_out")


def _linear_init(out_f, in_f, gen):
    """Same distribution as torch.nn.Linear's default init (kaiming_uniform a=sqrt(5) ->
    U(-1/sqrt(in), 1/sqrt(in)) for weight and bias); values are only used as test inputs."""
    bound = 1.0 / np.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * bound
    return w.float(), b.float()


def init_actor(obs, goal, action, seed):
    gen = torch.Generator().manual_seed(seed)
    dims = [(HIDDEN, obs + goal), (HIDDEN, HIDDEN), (HIDDEN, HIDDEN), (action, HIDDEN)]
    p = {}
    for k, (o, i) in zip(ACTOR_KEYS, dims):
        p[k + ".weight"], p[k + ".bias"] = _linear_init(o, i, gen)
    return p


def init_critic(obs, goal, action, seed):
    gen = torch.Generator().manual_seed(seed)
    dims = [(HIDDEN, obs + goal + action), (HIDDEN, HIDDEN), (HIDDEN, HIDDEN), (1, HIDDEN)]
    p = {}
    for k, (o, i) in zip(CRITIC_KEYS, dims):
        p[k + ".weight"], p[k + ".bias"] = _linear_init(o, i, gen)
    return p


def actor_forward(p, x, max_action):
    """models.py:19-26."""
    h = F.relu(F.linear(x, p["fc1.weight"], p["fc1.bias"]))
    h = F.relu(F.linear(h, p["fc2.weight"], p["fc2.bias"]))
    h = F.relu(F.linear(h, p["fc3.weight"], p["fc3.bias"]))
    return max_action * torch.tanh(F.linear(h, p["action_out.weight"], p["action_out.bias"]))


def critic_forward(p, x, actions, max_action):
    """models.py:37-44."""
    h = torch.cat([x, actions / max_action], dim=1)
    h = F.relu(F.linear(h, p["fc1.weight"], p["fc1.bias"]))
    h = F.relu(F.linear(h, p["fc2.weight"], p["fc2.bias"]))
    h = F.relu(F.linear(h, p["fc3.weight"], p["fc3.bias"]))
    return F.linear(h, p["q_out.weight"], p["q_out.bias"])


def flatten(tensors):
    """utils.py:18-27 / 60-69: named_parameters order, C-order flatten, float32."""
    return np.concatenate([t.detach().cpu().numpy().ravel() for t in tensors]).astype(np.float32)


class DDPGLearner:
    """State + one-step update of ddpg_agent.py (learner half only)."""

    def __init__(self, actor, critic, max_action=0.5, gamma=0.98, action_l2=1.0,
                 lr_actor=1e-3, lr_critic=1e-3, polyak=0.95, allreduce_sum=None):
        self.max_action, self.gamma, self.action_l2, self.polyak = max_action, gamma, action_l2, polyak
        self.actor = {k: v.clone().requires_grad_(True) for k, v in actor.items()}
        self.critic = {k: v.clone().requires_grad_(True) for k, v in critic.items()}
        self.actor_target = {k: v.clone() for k, v in actor.items()}        # ddpg_agent.py:30-34
        self.critic_target = {k: v.clone() for k, v in critic.items()}
        self.actor_optim = torch.optim.Adam(list(self.actor.values()), lr=lr_actor)     # :42
        self.critic_optim = torch.optim.Adam(list(self.critic.values()), lr=lr_critic)  # :43
        self._allreduce_sum = allreduce_sum      # utils.py:43-48 (SUM, not mean); None = 1 rank
        self.last = {}

    def _sync_grads(self, params):
        if self._allreduce_sum is None:
            return
        flat = self._allreduce_sum(flatten([p.grad for p in params.values()]))
        off = 0
        for p in params.values():
            n = p.numel()
            p.grad.copy_(torch.from_numpy(flat[off:off + n].reshape(tuple(p.shape))))
            off += n

    def update(self, x, x_next, actions, r):
        """ddpg_agent.py:250-277 on an already normalised minibatch (float32 tensors)."""
        with torch.no_grad():
            a_next = actor_forward(self.actor_target, x_next, self.max_action)
            q_next = critic_forward(self.critic_target, x_next, a_next, self.max_action)
            target_q = r + self.gamma * q_next
            target_q = torch.clamp(target_q, -1 / (1 - self.gamma), 0)          # :259-260
        real_q = critic_forward(self.critic, x, actions, self.max_action)
        critic_loss = (target_q - real_q).pow(2).mean()
        a_real = actor_forward(self.actor, x, self.max_action)
        actor_loss = -critic_forward(self.critic, x, a_real, self.max_action).mean()
        actor_loss = actor_loss + self.action_l2 * (a_real / self.max_action).pow(2).mean()
        self.actor_optim.zero_grad()
        actor_loss.backward()
        self._sync_grads(self.actor)
        actor_grads = flatten([p.grad for p in self.actor.values()])
        self.actor_optim.step()
        self.critic_optim.zero_grad()
        critic_loss.backward()
        self._sync_grads(self.critic)
        critic_grads = flatten([p.grad for p in self.critic.values()])
        self.critic_optim.step()
        self.last = dict(actor_loss=float(actor_loss.detach()), critic_loss=float(critic_loss.detach()),
                         actor_grads=actor_grads, critic_grads=critic_grads)
        return self.last

    def soft_update(self):
        """ddpg_agent.py:220-222, both nets (called once per cycle, :149-150)."""
        with torch.no_grad():
            for tgt, src in ((self.actor_target, self.actor), (self.critic_target, self.critic)):
                for k in tgt:
                    tgt[k].copy_((1 - self.polyak) * src[k].data + self.polyak * tgt[k].data)

    def flat(self, which):
        return flatten(list(getattr(self, which).values()))


def minibatch_tensors(transitions, o_norm, g_norm, clip_obs=200):
    """ddpg_agent.py:228-243: clip, normalise, concatenate, cast to float32 tensors."""
    from .running_norm import preproc_og

    o, g = preproc_og(transitions["obs"], transitions["g"], clip_obs)
    o_next, g_next = preproc_og(transitions["obs_next"], transitions["g"], clip_obs)
    x = np.concatenate([o_norm.normalize(o), g_norm.normalize(g)], axis=1)
    x_next = np.concatenate([o_norm.normalize(o_next), g_norm.normalize(g_next)], axis=1)
    return (torch.tensor(x, dtype=torch.float32), torch.tensor(x_next, dtype=torch.float32),
            torch.tensor(transitions["actions"], dtype=torch.float32),
            torch.tensor(transitions["r"], dtype=torch.float32))
