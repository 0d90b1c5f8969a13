"""Race check: N training cycles twice with gather-ahead on, once with it off -> all three must end in identical bits
(actor, critic, target critic, normalizer stats, RNG state)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.arguments import Args
from rl_arm_under_sparse_reward_amd.ddpg_agent import ddpg_agent, NET_ACTOR, NET_CRITIC, NET_CRITIC_TARGET
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes

def run(cycles, batch):
    ctx = _lib.Context.default()
    rng = DeviceRandomState(125, ctx=ctx)
    torch.manual_seed(0)
    ag = ddpg_agent(Args(batch_size=batch, buffer_size=64 * 100), None, dict(ENV_PARAMS), ctx=ctx, rng=rng)
    ag.buffer.store_episode(make_episodes(64, seed=1))
    pool = [make_episodes(2, seed=100 + i) for i in range(8)]
    for c in range(cycles):
        ag.train_cycle(pool[c % 8], 40)
    ctx.synchronize()
    import ctypes as C
    word = C.c_uint32()
    _lib.check(ag.lib.hp_agent_status(ag.h, C.byref(word)))   # sticky fault word of the in-launch hand-offs (k_cycle_open, k_fb_split8)
    assert word.value == 0, hex(word.value)
    h = hashlib.sha256()
    for slot in (NET_ACTOR, NET_CRITIC, NET_CRITIC_TARGET):
        h.update(ag._get_flat(slot).tobytes())
    h.update(np.asarray(ag.o_norm.mean).tobytes()); h.update(np.asarray(ag.g_norm.std).tobytes())
    st = rng.get_state(); h.update(np.asarray(st[1]).tobytes()); h.update(str(st[2]).encode())
    return h.hexdigest(), ag.last_losses(1)[0]

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for batch in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("256", "1024"))]:
    a = run(cycles, batch); b = run(cycles, batch)
    os.environ["RLARM_UPDATE_GRAPH"] = "0"; c = run(cycles, batch); del os.environ["RLARM_UPDATE_GRAPH"]
    # the split launch (in-launch counters between chains and tiles, slab8_split.h) against the two-launch form
    os.environ["RLARM_SPLIT"] = "0"; d = run(cycles, batch); del os.environ["RLARM_SPLIT"]
    print(f"batch {batch}: {cycles} cycles = {40 * cycles} updates  run1 {a[0][:16]} run2 {b[0][:16]} eager {c[0][:16]} "
          f"two-launch form {d[0][:16]}  losses {a[1]}", "OK" if a[0] == b[0] == c[0] == d[0] else "MISMATCH")
