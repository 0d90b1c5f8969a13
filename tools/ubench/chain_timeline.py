"""Stage stamps of chain slab 0 (hp_agent_debug_timeline; time-line build) for any batch: actor-side chain, critic-side chain.
RLARM_LIB=.../librlarm_hip_tl.so BATCH=1024 python tools/ubench/chain_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.replay_buffer import DeviceEpisodeBuffer
from rl_arm_under_sparse_reward_amd.normalizer import normalizer
from rl_arm_under_sparse_reward_amd.her import squared_threshold
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes
ctx = _lib.Context(0); lib = ctx.lib
B = int(os.environ.get("BATCH", "1024"))
rng = DeviceRandomState(125, ctx=ctx)
buf = DeviceEpisodeBuffer(5000, 100, 27, 3, 4, ctx=ctx)
buf.store(rng, make_episodes(5000, seed=1))
on, gn = normalizer(27, default_clip_range=5, ctx=ctx), normalizer(3, default_clip_range=5, ctx=ctx)
cfg = _lib.AgentCfg(obs_dim=27, goal_dim=3, act_dim=4, hidden=256, batch=B, grad_world_size=1, max_action=0.5, gamma=0.98,
                    action_l2=1.0, lr_actor=1e-3, lr_critic=1e-3, polyak=0.95, clip_obs=200.0, clip_range=5.0,
                    adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8)
h = C.c_void_p(); _lib.check(lib.hp_agent_create(ctx.h, C.byref(cfg), C.byref(h)))
rs = np.random.RandomState(0)
for net, n in ((0, 140548), (1, 140801), (2, 140548), (3, 140801)):
    _lib.check(lib.hp_agent_set_params(h, net, _lib.ptr((rs.uniform(-0.06, 0.06, n)).astype(np.float32), C.c_float), n))
for _ in range(10):
    _lib.check(lib.hp_agent_sample_and_update(h, buf.h, on.h, gn.h, rng.h, 0.8, squared_threshold(0.05), int(os.environ.get("SEQ", "6"))))
ctx.synchronize()
tl = (C.c_uint64 * 192)(); _lib.check(lib.hp_agent_debug_timeline(h, tl))
for ch in range(6):
    v = [tl[ch * 32 + k] for k in range(32)]
    if v[0]:
        pts = [(k, (v[k] - v[0]) / 100) for k in range(32) if v[k] and v[k] >= v[0]]
        print(f"[batch {B} channel {ch}]", " ".join(f"{k}:{t:.1f}" for k, t in pts))
        print("      steps:", " ".join(f"{b[0]}:{b[1] - a[1]:.2f}" for a, b in zip(pts, pts[1:])))
