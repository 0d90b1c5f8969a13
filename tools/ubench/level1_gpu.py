"""INTEGRATION.md section 1 with a GPU learner: the reference's torch update (ddpg_agent.py:250-277, autograd + torch.optim.Adam) on
cuda:0, fed (a) by replay_buffer.sample_device() -- hp_buffer_sample_dev, nothing crosses PCIe -- and (b) by the host-output path
the reference code would use unchanged: sample() -> _preproc_og -> normalize() x 4 -> np.concatenate -> torch.tensor(...).cuda()
(ddpg_agent.py:227-248).  Prints us per update of both and of their parts.  Measurement helper, not product code."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import ddpg_update as oupd
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.her import her_sampler
from rl_arm_under_sparse_reward_amd.normalizer import normalizer
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.replay_buffer import replay_buffer
from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes

B = int(os.environ.get("BATCH", "256"))
ctx = _lib.Context(0)
dev = torch.device("cuda", 0)
rng = DeviceRandomState(125, ctx=ctx)
her = her_sampler("future", 4, None, rng=rng)
buf = replay_buffer(dict(ENV_PARAMS), 5000 * 100, her.sample_her_transitions, rng=rng, ctx=ctx)
buf.store_episode(make_episodes(5000, seed=1))
o_norm, g_norm = normalizer(27, default_clip_range=5, ctx=ctx), normalizer(3, default_clip_range=5, ctx=ctx)
eps = make_episodes(2, seed=3)
o_norm.update(eps[0][:, :100].reshape(-1, 27)); g_norm.update(eps[2].reshape(-1, 3))
o_norm.recompute_stats(); g_norm.recompute_stats()

actor = {k: v.to(dev).requires_grad_(True) for k, v in oupd.init_actor(27, 3, 4, 1).items()}
critic = {k: v.to(dev).requires_grad_(True) for k, v in oupd.init_critic(27, 3, 4, 2).items()}
actor_t = {k: v.detach().clone() for k, v in actor.items()}
critic_t = {k: v.detach().clone() for k, v in critic.items()}
opt_a, opt_c = torch.optim.Adam(list(actor.values()), lr=1e-3), torch.optim.Adam(list(critic.values()), lr=1e-3)


def update(x, xn, a, r):     # ddpg_agent.py:250-277
    with torch.no_grad():
        q_next = oupd.critic_forward(critic_t, xn, oupd.actor_forward(actor_t, xn, 0.5), 0.5)
        y = torch.clamp(r + 0.98 * q_next, -50.0, 0)
    critic_loss = (y - oupd.critic_forward(critic, x, a, 0.5)).pow(2).mean()
    pi = oupd.actor_forward(actor, x, 0.5)
    actor_loss = -oupd.critic_forward(critic, x, pi, 0.5).mean() + (pi / 0.5).pow(2).mean()
    opt_a.zero_grad(); actor_loss.backward(); opt_a.step()
    opt_c.zero_grad(); critic_loss.backward(); opt_c.step()


def dev_path():
    mb = buf.sample_device(B, o_norm, g_norm, clip_obs=200)
    return mb["x"], mb["x_next"], mb["actions"], mb["r"]


def host_path():             # ddpg_agent.py:227-248 as the reference writes it, on the mirror's host-output objects
    tr = buf.sample(B)
    o, g, on_ = np.clip(tr['obs'], -200, 200), np.clip(tr['g'], -200, 200), np.clip(tr['obs_next'], -200, 200)
    x = np.concatenate([o_norm.normalize(o), g_norm.normalize(g)], axis=1)
    xn = np.concatenate([o_norm.normalize(on_), g_norm.normalize(g)], axis=1)
    return (torch.tensor(x, dtype=torch.float32).cuda(), torch.tensor(xn, dtype=torch.float32).cuda(),
            torch.tensor(tr['actions'], dtype=torch.float32).cuda(), torch.tensor(tr['r'], dtype=torch.float32).cuda())


def timed(fn, n):
    torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); ctx.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


for _ in range(20):
    update(*dev_path())
mb = dev_path()
print(f"batch {B}, 5000-episode shard, torch {torch.__version__} learner on {torch.cuda.get_device_name(0)}")
print(f"  sample_device() alone                       {timed(dev_path, 400):8.1f} us per minibatch (index draw + fused gather, asynchronous)")
print(f"  host-output path alone (sample + 4 normalize + H2D) {timed(host_path, 100):8.1f} us per minibatch")
print(f"  torch update alone (same minibatch)         {timed(lambda: update(*mb), 200):8.1f} us per update")
print(f"  sample_device() + torch update              {timed(lambda: update(*dev_path()), 200):8.1f} us per update")
print(f"  host-output path + torch update             {timed(lambda: update(*host_path()), 100):8.1f} us per update")
