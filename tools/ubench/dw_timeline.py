"""Per-workgroup time line of the weight-gradient launch (k_gemm_lds_adam[_ride][_u], gemm_lds.h) of the LAST update run.
Needs the time-line build: make -C rl_arm_under_sparse_reward_amd/csrc timeline;
RLARM_LIB=.../librlarm_hip_tl.so BATCH=512 python tools/ubench/dw_timeline.py"""
import ctypes as C, os, statistics as st, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.replay_buffer import DeviceEpisodeBuffer
from rl_arm_under_sparse_reward_amd.normalizer import normalizer
from rl_arm_under_sparse_reward_amd.her import squared_threshold
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes
ctx = _lib.Context(0); lib = ctx.lib
B = int(os.environ.get("BATCH", "512"))
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
    w = (rs.uniform(-0.06, 0.06, n)).astype(np.float32)
    _lib.check(lib.hp_agent_set_params(h, net, _lib.ptr(w, C.c_float), n))
n_up = int(os.environ.get("SEQ", "6"))
for _ in range(20):
    _lib.check(lib.hp_agent_sample_and_update(h, buf.h, on.h, gn.h, rng.h, 0.8, squared_threshold(0.05), n_up))
ctx.synchronize()
fn = lib._cdll.hp_debug_gemm_wg_timeline
wg = (C.c_uint64 * 4096)(); fn.restype = C.c_int; fn(wg)
rows = [(b, [wg[8 * b + k] for k in range(8)]) for b in range(512) if wg[8 * b] and wg[8 * b + 5]]
last = max(r[1][0] for r in rows)
rows = [r for r in rows if r[1][0] > last - 3000]        # the last launch only (stamps of earlier, larger launches may linger)
b0 = min(r[1][0] for r in rows)
def stat(v): return f"n={len(v):3d} min {min(v):6.2f} med {st.median(v):6.2f} max {max(v):6.2f}"
print(f"batch {B}: {len(rows)} stamped tile workgroups of the last launch")
for k, kn in ((0, "start"), (5, "end")):
    print(f"  since the launch's first start: {kn:6s}", stat([(r[1][k] - b0) / 100 for r in rows]))
for k, kn in ((1, "products done"), (3, "LDS sums ready"), (4, "gate + bias step done"), (5, "end")):
    print(f"  since own start: {kn:22s}", stat([(r[1][k] - r[1][0]) / 100 for r in rows if r[1][k]]))
slow = sorted(((r[1][5] - b0) / 100, (r[1][0] - b0) / 100, r[0]) for r in rows)[-12:]
print("  last to end (end, start, block):", " ".join(f"{e:.2f}<-{s:.2f}@{b}" for e, s, b in slow))
late = sorted(((r[1][0] - b0) / 100, r[0]) for r in rows)[-8:]
print("  last to start (start, block):", " ".join(f"{s:.2f}@{b}" for s, b in late))
riders = [r for r in rows if r[1][7] in (1, 2) and not r[1][1]]
for kind, nm in ((1, "index-plan rider"), (2, "gather riders")):
    v = [r for r in riders if r[1][7] == kind]
    if v: print(f"  {nm}: n={len(v)} start {min((r[1][0] - b0) / 100 for r in v):.2f} .. end {max((r[1][5] - b0) / 100 for r in v):.2f} (longest {max((r[1][5] - r[1][0]) / 100 for r in v):.2f} us)")
tiles = [r for r in rows if r[1][1]]
print("  tiles only: end", stat([(r[1][5] - b0) / 100 for r in tiles]))
if os.environ.get("TL_ROWS"):
    print("  slowest 30 workgroups (block: start | products, loop left, LDS sums, gate + bias, end since own start | end):")
    for r in sorted(tiles, key=lambda r: r[1][5])[-30:]:
        print(f"   wg {r[0]:3d} xcd {r[0] % 8}: start {(r[1][0] - b0) / 100:5.2f} | " + " ".join(f"{(r[1][k] - r[1][0]) / 100:6.2f}" if r[1][k] else "   -  " for k in (1, 2, 3, 4, 5)) + f" | end {(r[1][5] - b0) / 100:5.2f}")
