"""Soak for the in-launch hand-offs of round 3: many training cycles with the opening work as ONE launch (k_cycle_open: slots ->
scatter, normalizer plan -> normalizer update behind flags) against the same cycles with the four separate launches, and -- for the
split weight-gradient tiles (tickets) -- the same run twice.  Everything must agree bit for bit: parameters, targets, normalizer
statistics, the random stream, the buffer.  Usage: python tools/debug/soak_cycle_open.py [cycles] [batch]"""
import os
import sys
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

from rl_arm_under_sparse_reward_amd.arguments import Args
from rl_arm_under_sparse_reward_amd.ddpg_agent import NET_ACTOR, NET_ACTOR_TARGET, NET_CRITIC, NET_CRITIC_TARGET, ddpg_agent
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256


def run(env):
    for k, v in env.items():
        os.environ[k] = v
    torch.manual_seed(0)
    rng = DeviceRandomState(7)
    ag = ddpg_agent(Args(batch_size=batch, buffer_size=200 * 100), None, dict(ENV_PARAMS), rng=rng)
    ag.buffer.store_episode(make_episodes(150, seed=1, mode="walk"))
    pool = [make_episodes(2, seed=100 + i, mode="walk") for i in range(16)]
    for c in range(cycles):
        ag.train_cycle(pool[c % 16])        # the buffer (200 episodes) overflows after 25 cycles: random slots from then on
    crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())
    out = [crc(ag._get_flat(n)) for n in (NET_ACTOR, NET_CRITIC, NET_ACTOR_TARGET, NET_CRITIC_TARGET)]
    out += [crc(ag.o_norm.mean), crc(ag.o_norm.std), crc(ag.g_norm.mean), crc(ag.g_norm.std)]
    st = rng.get_state()
    out += [crc(st[1]), int(st[2]), crc(ag.buffer.buffers["obs"]), crc(ag.buffer.buffers["g"]), crc(ag.last_losses(40))]
    for k in env:
        os.environ.pop(k, None)
    return out


a = run({"RLARM_CYCLE_OPEN": "1"})
b = run({"RLARM_CYCLE_OPEN": "0"})
c = run({"RLARM_CYCLE_OPEN": "1"})
print(f"{cycles} cycles at batch {batch}: one-launch opening vs four launches: {'identical' if a == b else 'DIFFERENT'}; "
      f"same run twice: {'identical' if a == c else 'DIFFERENT'}")
if a != b or a != c:
    print(a, b, c, sep="\n")
    sys.exit(1)
