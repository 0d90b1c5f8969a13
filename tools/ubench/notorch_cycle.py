"""Time the cycle graph from a process that never imports torch (isolates runtime/tool overheads)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    print("torch imported", torch.__version__, "cuda init:", torch.cuda.is_available())
    if len(sys.argv) > 2: torch.zeros(1, device="cuda"); print("torch cuda context created")
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.replay_buffer import DeviceEpisodeBuffer
from rl_arm_under_sparse_reward_amd.normalizer import normalizer
from rl_arm_under_sparse_reward_amd.her import squared_threshold
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes
ctx = _lib.Context(0); lib = ctx.lib
def clock(tag):
    v = C.c_double(); _lib.check(lib.hp_ctx_clock_mhz(ctx.h, C.byref(v))); print(f"[clock {tag}] {v.value:.0f} MHz")
def floor(tag):
    clock(tag)
    for g in (1, 0):
        v = C.c_double(); _lib.check(lib.hp_ctx_launch_floor(ctx.h, 400, g, C.byref(v)))
        print(f"[floor {tag}] {'graph' if g else 'eager'}: {v.value:.2f} us/kernel")
floor("after ctx")
rng = DeviceRandomState(125, ctx=ctx)
buf = DeviceEpisodeBuffer(5000, 100, 27, 3, 4, ctx=ctx)
buf.store(rng, make_episodes(5000, seed=1))
floor("after 150MB buffer + store")
on, gn = normalizer(27, default_clip_range=5, ctx=ctx), normalizer(3, default_clip_range=5, ctx=ctx)
cfg = _lib.AgentCfg(obs_dim=27, goal_dim=3, act_dim=4, hidden=256, batch=int(os.environ.get("BATCH", "256")), grad_world_size=1, max_action=0.5, gamma=0.98,
                    action_l2=1.0, lr_actor=1e-3, lr_critic=1e-3, polyak=0.95, clip_obs=200.0, clip_range=5.0,
                    adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8)
h = C.c_void_p(); _lib.check(lib.hp_agent_create(ctx.h, C.byref(cfg), C.byref(h)))
floor("after agent create")
rs = np.random.RandomState(0)
for net, n in ((0, 140548), (1, 140801), (2, 140548), (3, 140801)):
    w = (rs.uniform(-0.06, 0.06, n)).astype(np.float32)
    _lib.check(lib.hp_agent_set_params(h, net, _lib.ptr(w, C.c_float), n))
eps = make_episodes(2, seed=3)
d = C.c_double
def chain():
    names = {6: "adam", 8: "polyak", 10: "forward+backward (active engine)", 11: "chain kernel", 12: "weight gradients + optimizer"}
    for k, nm in names.items():
        v = C.c_double(); _lib.check(lib.hp_agent_debug_chain(h, k, 200, C.byref(v))); print(f"[chain] {nm}: {v.value:.2f} us/launch")
chain()
def cycle(nb):
    _lib.check(lib.hp_agent_train_cycle(h, buf.h, on.h, gn.h, rng.h, *[_lib.ptr(a, d) for a in eps], 2, 0.8, squared_threshold(0.05), nb))
for nb in (40,):
    for _ in range(5): cycle(nb)
    ctx.synchronize()
    reps = max(1, 2000 // nb)
    t0 = time.perf_counter()
    for _ in range(reps): cycle(nb)
    ctx.synchronize()
    clock(f"right after {reps} cycles, before sync")
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"n_batches={nb}: {1e6*dt/(reps*nb):.1f} us/step  ({1e6*dt/reps:.0f} us/cycle)")
tl = (C.c_uint64 * 192)(); _lib.check(lib.hp_agent_debug_timeline(h, tl))
for ch, nm in ((0, "fwd T | merged critic side"), (1, "fwd A | merged actor side")):
    v = [tl[ch * 32 + k] for k in range(32)]
    if v[0]:
        print(f"[timeline {nm}]", " ".join(f"{k}:{(v[k]-v[0])/100:.1f}" for k in range(32) if v[k]))
for base, nm in ((160, "dW gemm first wg"), (176, "dW gemm last wg")):
    v = [tl[base + k] for k in range(8)]
    if v[0]:
        print(f"[timeline {nm}]", " ".join(f"{k}:{(v[k]-tl[160])/100:.1f}" for k in range(8) if v[k]),
              "| first stamp at", (tl[base] - tl[0]) / 100 if tl[0] else None, "us after chain start")
if os.environ.get("RLARM_ENGINE") == "slab32" or int(os.environ.get("BATCH", "256")) > 2048:
    base = tl[96]
    for w in range(8):
        print(f"[timeline slab32 layer wave {w}] " + " ".join(f"{k}:{(tl[96 + 4 * w + k] - base) / 100:.2f}" for k in range(4)) + f" after-sync:{(tl[128 + 4 * w] - base) / 100:.2f}")
if tl[160 + 24] and tl[160 + 25] and tl[161] > tl[160]:
    print(f"[timeline dW] shader clock over workgroup 0's product loop: {100.0 * (tl[160 + 25] - tl[160 + 24]) / (tl[161] - tl[160]):.0f} MHz")
for k, nm in ((9, "K loop"), (11, "ticket"), (13, "end")):
    if tl[160 + k]: print(f"[timeline dW latest {nm}] {((tl[160 + k] >> 12) - tl[160]) / 100:.1f} us by workgroup {tl[160 + k] & 4095}")
if hasattr(lib, "hp_debug_gemm_wg_timeline") or True:
    try:
        fn = lib.hp_debug_gemm_wg_timeline        # time-line builds only
        wg = (C.c_uint64 * 4096)(); fn.restype = C.c_int; fn(wg)
        t0s = [wg[8 * b] for b in range(512) if wg[8 * b]]
        if t0s:
            base = min(t0s)
            import statistics as st
            def col(k, sel): return [(wg[8 * b + k] - base) / 100 for b in sel if wg[8 * b + k] and wg[8 * b]]
            groups = (("256 x 256 tiles (wg 0-255)", range(0, 256)), ("narrow tiles (wg 256+)", range(256, 512)))
            for nm, sel in groups:
                for k, kn in ((0, "start"), (1, "products done"), (3, "LDS sums ready"), (4, "exchange / sums done"), (5, "end")):
                    c = col(k, sel)
                    if c: print(f"[dW per-workgroup] {nm:28s} {kn:20s} n={len(c):3d} min {min(c):6.2f} median {st.median(c):6.2f} max {max(c):6.2f}")
    except AttributeError:
        pass
try:
    fn = lib.hp_debug_gemm_blk_timeline
    bk = (C.c_uint64 * 512)(); fn.restype = C.c_int; fn(bk)
    base = min(v for v in bk if v) if any(bk) else 0
    for w in range(8):
        row = [(bk[(w * 32 + i) * 2] , bk[(w * 32 + i) * 2 + 1]) for i in range(32)]
        if row[0][0]:
            print(f"[dW block loop wave {w}] " + " ".join(f"{(a - base) / 100:.2f}/{(b - base) / 100:.2f}" for a, b in row if a))
except AttributeError:
    pass
floor("after cycles")
# eager path
_lib.check(lib.hp_agent_sample_and_update(h, buf.h, on.h, gn.h, rng.h, 0.8, squared_threshold(0.05), 40)); ctx.synchronize()
t0 = time.perf_counter()
for _ in range(25): _lib.check(lib.hp_agent_sample_and_update(h, buf.h, on.h, gn.h, rng.h, 0.8, squared_threshold(0.05), 40))
ctx.synchronize(); dt = time.perf_counter() - t0
print(f"eager: {1e6*dt/1000:.1f} us/step")
