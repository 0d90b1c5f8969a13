"""Stand-alone timing of the device-output fused sampler (hp_buffer_sample_dev: k_draw_plan + k_gather_fused) over shard sizes and
batches: us per launch and GB/s by SURVEY 8d's 528 B / transition and by the 812 B this build moves.  Run under rocprofv3
(--kernel-trace --stats / --pmc FETCH_SIZE / WRITE_SIZE) for the per-kernel traffic.  Measurement helper, not product code.
  EPISODES=5000,10000,20000 BATCHES=256,262144 REPS=20 python tools/ubench/sample_fused.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.her import her_sampler
from rl_arm_under_sparse_reward_amd.normalizer import normalizer
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.replay_buffer import replay_buffer
from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes

ctx = _lib.Context(0)
reps = int(os.environ.get("REPS", "20"))
out = []
for n_eps in [int(x) for x in os.environ.get("EPISODES", "5000").split(",")]:
    rng = DeviceRandomState(125, ctx=ctx)
    her = her_sampler("future", 4, None, rng=rng)
    buf = replay_buffer(dict(ENV_PARAMS), n_eps * 100, her.sample_her_transitions, rng=rng, ctx=ctx)
    base = make_episodes(min(n_eps, 5000), seed=1)
    for lo in range(0, n_eps, 5000):      # (distinct contents are not needed for timing: the slots are what spreads the reads)
        n = min(5000, n_eps - lo)
        buf.store_episode([a[:n] for a in base])
    o_norm, g_norm = normalizer(27, default_clip_range=5, ctx=ctx), normalizer(3, default_clip_range=5, ctx=ctx)
    eps = make_episodes(2, seed=3)
    o_norm.update(eps[0][:, :100].reshape(-1, 27)); g_norm.update(eps[2].reshape(-1, 3))
    o_norm.recompute_stats(); g_norm.recompute_stats()
    shard_mb = n_eps * 29840 / 1e6
    f32 = os.environ.get("F32_ROWS", "0") == "1"      # the throughput rows (hp_buffer_enable_f32_rows / hp_buffer_sample_dev_f32)
    if f32:
        _lib.check(ctx.lib.hp_buffer_enable_f32_rows(buf._dev.h))
    for nb in [int(x) for x in os.environ.get("BATCHES", "256,16384,262144").split(",")]:
        d, g = C.c_double(), C.c_double()
        _lib.check(ctx.lib.hp_buffer_sample_dev_us(buf._dev.h, rng.h, o_norm.h, g_norm.h, nb, float(her.future_p),
                                                   float(her.sq_threshold), 200.0, reps if nb > 4096 else 10 * reps, 1 if f32 else 0,
                                                   C.byref(d), C.byref(g)))
        moved = 572 if f32 else 812        # f32 rows: 62 float32 + 6 float64 goals + the 16 B index record read, 65 float32 written
        rec = {"rows": "f32 mirror" if f32 else "f64", "episodes": n_eps, "shard_MB_f64": round(shard_mb, 1), "batch": nb,
               "gather_us": round(g.value, 3), "draw_us": round(d.value, 3),
               "GBps_528B": round(528 * nb / g.value / 1e3, 1), "GBps_moved": round(moved * nb / g.value / 1e3, 1), "bytes_moved_per_transition": moved,
               "G_transitions_per_s": round(nb / g.value / 1e3, 3)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    del buf
