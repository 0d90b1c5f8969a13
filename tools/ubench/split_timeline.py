"""Per-workgroup time line of the split launch (k_fb_split8, slab8_split.h) + the actor tile launch behind it.
Needs the time-line build: make -C rl_arm_under_sparse_reward_amd/csrc timeline; RLARM_LIB=.../librlarm_hip_tl.so python tools/ubench/split_timeline.py"""
import ctypes as C, os, statistics as st, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.replay_buffer import DeviceEpisodeBuffer
from rl_arm_under_sparse_reward_amd.normalizer import normalizer
from rl_arm_under_sparse_reward_amd.her import squared_threshold
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes
ctx = _lib.Context(0); lib = ctx.lib
B = int(os.environ.get("BATCH", "256"))
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
eps = make_episodes(2, seed=3)
d = C.c_double
def cycle(nb):
    _lib.check(lib.hp_agent_train_cycle(h, buf.h, on.h, gn.h, rng.h, *[_lib.ptr(a, d) for a in eps], 2, 0.8, squared_threshold(0.05), nb))
for _ in range(5): cycle(40)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(50): cycle(40)
ctx.synchronize()
print(f"{1e6 * (time.perf_counter() - t0) / 2000:.2f} us/update (time-line build)")
# one more sequence whose LAST split launch still has target chains: 39 updates + 1 -> stamps of the last launch of a 3-update call
_lib.check(lib.hp_agent_sample_and_update(h, buf.h, on.h, gn.h, rng.h, 0.8, squared_threshold(0.05), int(os.environ.get("SEQ", "6"))))
ctx.synchronize()
fn = lib._cdll.hp_debug_split_timeline
out = (C.c_uint64 * 5120)(); fn.restype = C.c_int; fn(out)
rows = [(b, [out[5 * b + 0], out[5 * b + 1], out[5 * b + 2], out[5 * b + 3], out[5 * b + 4] & 255, out[5 * b + 4] >> 8]) for b in range(1024) if out[5 * b]]
base = min(r[1][0] for r in rows)
names = ["A actor side", "C critic", "T target", "plan", "gather", "warm", "tile (critic dW + Adam)"]   # slab8_split_args.h: SR_*
TILE = 6
def stat(v): return f"n={len(v):3d} min {min(v):6.2f} med {st.median(v):6.2f} max {max(v):6.2f}" if v else "-"
rows = [r for r in rows if r[1][3] >= r[1][0]]
base = min(r[1][0] for r in rows if r[1][4] == 0)      # first actor-side chain's start
for role, nm in enumerate(names):
    sel = [r for r in rows if r[1][4] == role and r[1][0] >= base - 200]
    if not sel: continue
    for k, kn in ((0, "start"), (1, "hand-off point"), (2, "gate reached"), (5, "gate passed"), (3, "end")):
        if role != TILE and k == 5: continue
        if role not in (TILE, 0) and k == 2: continue
        v = [(r[1][k] - base) / 100 for r in sel if r[1][k]]
        if v: print(f"[split] {nm:26s} {kn:15s} {stat(v)}")
try:
    fe = lib._cdll.hp_debug_split_entry
    ent = (C.c_uint64 * 1024)(); fe.restype = C.c_int; fe(ent)
    for role, nm in ((0, "A"), (1, "C"), (2, "T"), (TILE, "tile")):
        v = [(r[1][0] - ent[r[0]]) / 100 for r in rows if r[1][4] == role and ent[r[0]] and r[1][0] >= ent[r[0]]]
        if v: print(f"[split] {nm}: first instruction -> role known (the first kernel-argument fetch)  {stat(v)}")
except AttributeError:
    pass
late = sorted((r[0], r[1][4], (r[1][0] - base) / 100, (r[1][3] - base) / 100) for r in rows if r[1][0] >= base - 200 and r[1][4] in (0, 1, 2) and (r[1][0] - base) / 100 > 1.0)
print("[split] workgroups of the first roles that started > 1 us late (block, XCD, slot, role, start, end):",
      " ".join(f"{b}:x{b % 8}s{b // 8}r{ro}:{t0:.1f}-{t1:.1f}" for b, ro, t0, t1 in late[:60]))
for role, nm in ((0, "A"), (1, "C"), (2, "T")):      # by XCD: is the spread of a role's end times a property of the XCD it runs on?
    for x in range(8):
        v = [(r[1][3] - base) / 100 for r in rows if r[1][4] == role and r[0] % 8 == x and r[1][0] >= base - 200]
        if v: print(f"[split by XCD] {nm} chains on XCD {x}: n={len(v):2d} end min {min(v):6.2f} med {st.median(v):6.2f} max {max(v):6.2f}")
tl = (C.c_uint64 * 192)(); _lib.check(lib.hp_agent_debug_timeline(h, tl))
for ch, nm in ((0, "A chain slab 0"), (3, "A chain, first slab of the next XCD"), (1, "C chain slab 0"), (2, "T chain slab 0")):
    v = [tl[ch * 32 + k] for k in range(32)]
    if v[0]: print(f"[timeline {nm}]", " ".join(f"{k}:{(v[k]-v[0])/100:.1f}" for k in range(32) if v[k]))
ad = [tl[168 + k] for k in range(4)]
if all(ad): print('[actor tile launch, workgroup 0] optimizer step: entered 0.00, math done %.2f, p / m / v stored %.2f, end (fragment copies stored) %.2f us' % tuple((ad[k] - ad[0]) / 100 for k in (1, 2, 3)))
# per-wave stamps of ONE 256 x 256 layer (the critic's first dX layer in the actor-side chain above; slab8.h: S8_WSTAMP)
wv = [[tl[128 + 8 * k + w] for w in range(8)] for k in range(4)]
if all(wv[0]):
    w0 = min(wv[0])
    print("[one layer, per wave] wave = (column group, reduction half); us since the first wave entered the layer")
    for k, kn in enumerate(("layer entered", "products done", "merge barrier passed", "closing barrier passed")):
        print(f"   {kn:24s}", " ".join(f"w{w}({w & 3},{w >> 2}):{(wv[k][w] - w0) / 100:5.2f}" for w in range(8) if wv[k][w]))
try:
    fn = lib._cdll.hp_debug_gemm_wg_timeline
    wg = (C.c_uint64 * 4096)(); fn.restype = C.c_int; fn(wg)
    # the actor launch behind the split launch has < 200 workgroups; the in-launch tiles of the split launch sit at higher indices
    first_tile = min(r[0] for r in rows if r[1][4] == TILE)
    for nm, sel in (("actor tile launch", range(0, min(first_tile, 200))), ("in-launch critic tiles (last launch)", range(first_tile, 512))):
        t0s = [wg[8 * b] for b in sel if wg[8 * b]]
        if not t0s: continue
        b0 = min(t0s)
        for k, kn in ((0, "start"), (5, "end")):
            c = [(wg[8 * b + k] - b0) / 100 for b in sel if wg[8 * b + k] and wg[8 * b]]
            if c: print(f"[{nm}] since the launch's first start: {kn:10s} {stat(c)}")
        slow = sorted(((wg[8 * b + 5] - b0) / 100, b) for b in sel if wg[8 * b + 5] and wg[8 * b])[-6:]
        print(f"[{nm}] last to end (us since first start, workgroup):", " ".join(f"{t:.2f}@{b}" for t, b in slow))
        for k, kn in ((0, "start"), (1, "products done"), (3, "LDS sums ready"), (4, "gate + bias step done"), (7, "optimizer step entered"), (5, "end")):
            c = [(wg[8 * b + k] - wg[8 * b]) / 100 for b in sel if wg[8 * b + k] and wg[8 * b]]
            if c: print(f"[{nm}] since own start: {kn:22s} {stat(c)}")
        if nm.startswith("actor") and os.environ.get("TL_ROWS", "1") != "0":
            # every workgroup of the launch behind the split launch: stamps since the launch's first start (0 start, 1 operands
            # landed / products done, 2 product loop left, 3 LDS sums ready, 4 gate + bias, 5 end)
            print(f"[{nm}] per workgroup (index: start | 1 2 3 4 5 since own start), slowest 40 by end:")
            per = sorted(((wg[8 * b + 5] - b0) / 100, b) for b in sel if wg[8 * b + 5] and wg[8 * b])
            for t, b in (per if os.environ.get('TL_ROWS') == 'all' else per[-40:]):
                hw = wg[8 * b + 6]
                print(f"   wg {b:3d} xcd {b % 8} slot {b // 8:2d} cu {(hw >> 8) & 15:2d} sh {(hw >> 12) & 1} se {(hw >> 13) & 7}: start {(wg[8 * b] - b0) / 100:5.2f} | " +
                      " ".join(f"{(wg[8 * b + k] - wg[8 * b]) / 100:5.2f}" if wg[8 * b + k] else "  -  " for k in (1, 2, 3, 4, 5)) + f" | end {t:5.2f}")
except AttributeError:
    pass
