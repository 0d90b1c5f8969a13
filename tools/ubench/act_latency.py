"""Host-side latency of one rollout policy call (hp_agent_act: H2D of the rows, normalise + actor, D2H, sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.arguments import Args
from rl_arm_under_sparse_reward_amd.ddpg_agent import ddpg_agent
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS
ctx = _lib.Context(0)
agent = ddpg_agent(Args(batch_size=256, buffer_size=10000), None, dict(ENV_PARAMS), ctx=ctx, rng=DeviceRandomState(1, ctx=ctx))
rs = np.random.RandomState(0)
for rows in (1, 16, 64, 256):
    obs, g = rs.normal(size=(rows, 27)), rs.normal(size=(rows, 3))
    for _ in range(20): agent.act(obs, g)
    t0 = time.perf_counter()
    for _ in range(500): agent.act(obs, g)
    print(f"engine={os.environ.get('RLARM_ENGINE', 'slab8')} rows={rows}: {1e6 * (time.perf_counter() - t0) / 500:.1f} us/call")
