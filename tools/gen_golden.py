#!/usr/bin/env python3
"""tools/gen_golden.py -- regenerate tests/golden/*.npz by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  Nothing here is
needed on the GPU box: the fixtures it writes are plain data (inputs + the reference's
outputs) and are committed.

How the reference is executed:
  * her.py, replay_buffer.py, models.py import as they are;
  * normalizer.py, utils.py, ddpg_agent.py import mpi4py (absent here) -> an in-memory
    stub module is installed first.  The stub supports W "ranks" as W threads of this
    process with a barrier-based summing Allreduce, which is how multi-rank fixtures
    (normalizer mean-of-ranks) are produced;
  * bmirobot_env/* cannot be imported (gym + pybullet absent).  The two functions of it
    that are on the hot path -- goal_distance (bmirobot_env_push_F.py:20-23) and
    compute_reward (:84-90) -- are pulled out of the reference source with `ast` and
    executed unmodified against a dummy object carrying reward_type/distance_threshold
    (bmirobot_push_F.py:9,20).
Every fixture is cross-checked against oracle/ before it is written, so a passing run
of this script is itself the "oracle == reference" proof on this container.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import ast
import contextlib
import io
import os
import sys
import tempfile
import threading
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------- mpi stub
class _World:
    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.slots = [None] * size
        self.local = threading.local()

    def rank(self):
        return getattr(self.local, "rank", 0)


_WORLD = _World(1)


class _Comm:
    def Get_rank(self):
        return _WORLD.rank()

    def Get_size(self):
        return _WORLD.size

    def _gather(self, x):
        w = _WORLD
        if w.size == 1:
            return [x]
        w.slots[w.rank()] = np.array(x, copy=True)
        w.barrier.wait()
        vals = [np.array(v, copy=True) for v in w.slots]
        w.barrier.wait()
        return vals

    def Allreduce(self, x, buf, op=None):
        vals = self._gather(x)
        acc = np.zeros_like(buf)
        for v in vals:                      # rank order 0..W-1, like a linear MPI_SUM
            acc = acc + v
        buf[...] = acc

    def allreduce(self, x, op=None):
        return sum(float(v) for v in self._gather(np.asarray(x)))

    def Bcast(self, x, root=0):
        vals = self._gather(x)
        x[...] = vals[root]


def install_mpi_stub():
    m = types.ModuleType("mpi4py")
    mpi = types.SimpleNamespace(COMM_WORLD=_Comm(), SUM="SUM")
    m.MPI = mpi
    sys.modules["mpi4py"] = m


def set_world(size):
    global _WORLD
    _WORLD = _World(size)


def run_ranks(fn, size):
    """Run fn(rank) on `size` stub ranks (threads); returns the list of results."""
    set_world(size)
    out, err = [None] * size, []

    def body(r):
        _WORLD.local.rank = r
        try:
            out[r] = fn(r)
        except BaseException as e:  # pragma: no cover
            err.append(e)
            _WORLD.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(size)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    set_world(1)
    if err:
        raise err[0]
    return out


# ----------------------------------------------------------------- reference extraction
def load_reference():
    install_mpi_stub()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.environ.setdefault("MPLBACKEND", "Agg")
    import her, replay_buffer, normalizer, models, utils, ddpg_agent, arguments  # noqa: E401

    return types.SimpleNamespace(her=her, replay_buffer=replay_buffer, normalizer=normalizer,
                                 models=models, utils=utils, ddpg_agent=ddpg_agent, arguments=arguments)


def reference_env():
    """Dummy env whose compute_reward is the reference's own function body."""
    path = os.path.join(REF, "bmirobot_env", "bmirobot_env_push_F.py")
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    ns = {"np": np}
    wanted = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "goal_distance":
            wanted["goal_distance"] = node
        if isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == "compute_reward":
                    wanted["compute_reward"] = sub
    mod = ast.Module(body=[wanted["goal_distance"], wanted["compute_reward"]], type_ignores=[])
    exec(compile(mod, path, "exec"), ns)

    class _Env:
        reward_type = "sparse"            # bmirobot_push_F.py:9
        distance_threshold = 0.05         # bmirobot_push_F.py:20
        compute_reward = ns["compute_reward"]

    return _Env()


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


ENV_PARAMS = {"obs": 27, "goal": 3, "action": 4, "action_max": 0.5, "max_timesteps": 100}


# ---------------------------------------------------------------------------- fixtures
def gen_rng_kat(ref, out):
    """F1: indices the reference's her.py derives from the legacy RNG stream."""
    from oracle.her_replay import draw_her_indices, future_probability

    env = reference_env()
    cases, payload = [], {}
    T = 100
    for seed in (125, 126):
        for n in (1, 2, 7, 64, 100, 5000):
            for B in (100, 256, 1024):
                for k in ((4, 8) if (n, B) == (100, 256) else (4,)):
                    # tagged episodes: outputs reveal (e, t, her, future_t)
                    ee, tt = np.meshgrid(np.arange(n), np.arange(T + 1), indexing="ij")
                    obs = np.stack([ee, tt], -1).astype(np.float64)
                    ag = np.stack([ee, tt, np.full_like(ee, 7)], -1).astype(np.float64)
                    g = -np.ones((n, T, 3))
                    act = np.zeros((n, T, 1))
                    batch = {"obs": obs, "ag": ag, "g": g, "actions": act,
                             "obs_next": obs[:, 1:], "ag_next": ag[:, 1:]}
                    sampler = ref.her.her_sampler("future", k, env.compute_reward.__get__(env))
                    np.random.seed(seed)
                    tr = sampler.sample_her_transitions(batch, B)
                    key, pos = np.random.get_state()[1:3]
                    e = tr["obs"][:, 0].astype(np.int64)
                    t = tr["obs"][:, 1].astype(np.int64)
                    her = tr["g"][:, 2] == 7
                    fut = np.where(her, tr["g"][:, 1], -1).astype(np.int64)
                    assert np.all(tr["g"][her, 0] == e[her])
                    # oracle cross-check
                    rs = np.random.RandomState(seed)
                    oe, ot, oh, of = draw_her_indices(rs, n, T, B, future_probability("future", k))
                    assert np.array_equal(oe, e) and np.array_equal(ot, t) and np.array_equal(oh, her)
                    assert np.array_equal(of[her], fut[her])
                    assert np.array_equal(rs.get_state()[1], key) and rs.get_state()[2] == pos
                    tag = f"s{seed}_n{n}_b{B}_k{k}"
                    cases.append(tag)
                    payload[tag + "_e"] = e.astype(np.int32)
                    payload[tag + "_t"] = t.astype(np.int16)
                    payload[tag + "_her"] = her
                    payload[tag + "_future_t"] = fut.astype(np.int16)
                    payload[tag + "_key"] = key.astype(np.uint32)
                    payload[tag + "_pos"] = np.int32(pos)
    payload["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(out, "rng_kat.npz"), **payload)
    print(f"rng_kat.npz: {len(cases)} cases")


def gen_her_samples(ref, out):
    """F2: full sample() outputs of the reference replay_buffer + her_sampler."""
    from oracle.her_replay import EpisodeStore, future_probability
    from rl_arm_under_sparse_reward_amd.synthetic import episode_checksum, make_episodes

    env = reference_env()
    payload, cases = {}, []
    for (n, B, k, mode, seed, dseed) in [(7, 256, 4, "walk", 125, 11), (64, 256, 8, "walk", 126, 12),
                                         (100, 1024, 4, "iid", 125, 1), (5, 100, 4, "walk", 3, 13),
                                         (1, 64, 4, "walk", 9, 14)]:
        eps = make_episodes(n, seed=dseed, mode=mode)
        sampler = ref.her.her_sampler("future", k, env.compute_reward.__get__(env))
        with quiet():
            buf = ref.replay_buffer.replay_buffer(ENV_PARAMS, n * 100, sampler.sample_her_transitions)
        buf.store_episode(eps)
        np.random.seed(seed)
        tr = buf.sample(B)
        key, pos = np.random.get_state()[1:3]
        # oracle cross-check (bitwise)
        st = EpisodeStore(100, 27, 3, 4, n * 100)
        rs = np.random.RandomState(seed)
        st.store_episode(eps, rs)
        otr, _ = st.sample(B, future_probability("future", k), rs)
        for kk in tr:
            assert tr[kk].dtype == otr[kk].dtype and np.array_equal(
                tr[kk].view(np.uint8), otr[kk].view(np.uint8)), kk
        tag = f"n{n}_b{B}_k{k}_{mode}"
        cases.append(tag)
        payload[tag + "_meta"] = np.array([n, B, k, seed, dseed], dtype=np.int64)
        payload[tag + "_mode"] = np.array(mode)
        payload[tag + "_checksum"] = np.float64(episode_checksum(eps))
        for kk, v in tr.items():
            payload[tag + "_" + kk] = v
        payload[tag + "_key"] = key.astype(np.uint32)
        payload[tag + "_pos"] = np.int32(pos)
        payload[tag + "_success_frac"] = np.float64(np.mean(tr["r"] == 0))
    payload["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(out, "her_sample.npz"), **payload)
    print("her_sample.npz:", cases, [float(payload[c + "_success_frac"]) for c in cases])


def gen_reward_adversarial(out):
    """F3: (ag, g) pairs around the 0.05 radius; rewards from the reference function."""
    import math

    from oracle.her_replay import compute_reward, squared_distance_threshold

    env = reference_env()
    rs = np.random.RandomState(5)
    M = 4096
    g = rs.uniform(-1, 1, (M, 3))
    u = rs.normal(size=(M, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    radius = np.full(M, 0.05)
    # nudge the radius by -6..+6 ulps of 0.05, plus exact zeros, tiny and huge offsets
    ulps = rs.randint(-6, 7, M)
    for i in range(M):
        r = 0.05
        for _ in range(abs(int(ulps[i]))):
            r = math.nextafter(r, math.inf if ulps[i] > 0 else -math.inf)
        radius[i] = r
    ag = g + u * radius[:, None]
    ag[:64] = g[:64]                                    # identical rows -> d = 0 -> -0.0
    ag[64:128] = g[64:128] + 1e-300                      # denormal-scale differences
    ag[128:192] = g[128:192] + rs.uniform(-3, 3, (64, 3))  # far
    # single-axis exact-threshold cases: |dx| = 0.05 exactly representable offset
    ag[192:256] = g[192:256]
    ag[192:256, 0] = g[192:256, 0] + 0.05
    # refine a block so that the *computed* squared distance sits within a few ulps of s*
    s_star = squared_distance_threshold(0.05)
    for i in range(256, 1280):
        d = ag[i] - g[i]
        s = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
        for _ in range(60):
            if abs(s - s_star) <= 4 * math.ulp(s_star):
                break
            scale = math.sqrt(s_star / s)
            ag[i] = g[i] + (ag[i] - g[i]) * scale
            d = ag[i] - g[i]
            s = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
    r = env.compute_reward(ag, g, None)
    assert r.dtype == np.float32
    ro = compute_reward(ag, g)
    assert np.array_equal(r.view(np.uint32), ro.view(np.uint32))
    d = ag - g
    s = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    assert np.array_equal((s >= s_star), (r != 0)), "squared-threshold rule disagrees with the reference"
    bits = r.view(np.uint32)
    assert set(np.unique(bits)) <= {0x80000000, 0xBF800000}
    near = np.sum(np.abs(s - s_star) <= 8 * math.ulp(s_star))
    np.savez_compressed(os.path.join(out, "reward_adversarial.npz"), ag=ag, g=g, r_bits=bits,
                        s_star=np.float64(s_star))
    print(f"reward_adversarial.npz: {M} pairs, {int(np.sum(bits == 0x80000000))} successes, {int(near)} within 8 ulp of s*")


def gen_storage_idx(ref, out):
    """F4: slot sequences of replay_buffer._get_storage_idx across its three branches."""
    from oracle.her_replay import EpisodeStore

    payload, cases = {}, []
    small = {"obs": 2, "goal": 1, "action": 1, "action_max": 1.0, "max_timesteps": 3}
    for (size_eps, incs, seed) in [(5, [2, 2, 2, 2, 1, 3, 1, 5, 7], 0), (5, [2, 2, 2, 2], 1), (5, [2, 2, 2, 2], 2),
                                   (5, [2, 2, 2, 2], 3), (16, [5, 5, 5, 5, 1, 1, 16, 20], 42), (3, [1, 1, 1, 1, 1], 7),
                                   (100, [64, 64, 2, 2, 2], 125)]:
        with quiet():
            buf = ref.replay_buffer.replay_buffer(small, size_eps * 3, None)
        st = EpisodeStore(3, 2, 1, 1, size_eps * 3)
        np.random.seed(seed)
        rs = np.random.RandomState(seed)
        seq, sizes = [], []
        for inc in incs:
            idx = np.atleast_1d(buf._get_storage_idx(inc)).astype(np.int64)
            oidx = np.atleast_1d(st.storage_slots(inc, rs)).astype(np.int64)
            assert np.array_equal(idx, oidx) and buf.current_size == st.current_size
            seq.append(idx)
            sizes.append(buf.current_size)
        key, pos = np.random.get_state()[1:3]
        tag = f"size{size_eps}_seed{seed}_n{len(incs)}"
        cases.append(tag)
        payload[tag + "_size"] = np.int64(size_eps)
        payload[tag + "_seed"] = np.int64(seed)
        payload[tag + "_incs"] = np.array(incs, dtype=np.int64)
        payload[tag + "_slots"] = np.concatenate(seq)
        payload[tag + "_current_size"] = np.array(sizes, dtype=np.int64)
        payload[tag + "_key"] = key.astype(np.uint32)
        payload[tag + "_pos"] = np.int32(pos)
    payload["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(out, "storage_idx.npz"), **payload)
    print("storage_idx.npz:", cases)


def _norm_inputs(rank, step, size):
    rs = np.random.RandomState(1000 + 17 * rank + step)
    n = [100, 100, 37, 250, 1, 100][step % 6]
    scale = [1.0, 30.0, 1e-3, 250.0, 1.0, 5.0][step % 6]     # 250 exceeds clip_obs=200 on purpose
    return rs.normal(0.3 * (rank + 1), scale, size=(n, size))


def gen_normalizer(ref, out):
    """F5: float32 bit patterns of the running normalizer, world sizes 1, 2, 4 and 8 (stub ranks; the 8-GPU BASELINE
    configs 4 / 5 exchange their statistics exactly like this)."""
    from oracle.running_norm import RunningNorm

    payload = {}
    for world in (1, 2, 4, 8):
        for size in (27, 3):
            def body(rank, size=size):
                nz = ref.normalizer.normalizer(size=size, default_clip_range=5)
                hist = []
                for step in range(6):
                    v = np.clip(_norm_inputs(rank, step, size), -200, 200)
                    nz.update(v)
                    if step % 2 == 1 or step == 4:          # sometimes two updates per recompute
                        nz.recompute_stats()
                        hist.append([np.array(a, copy=True) for a in
                                     (nz.mean, nz.std, nz.total_sum, nz.total_sumsq, nz.total_count)])
                probe = np.random.RandomState(5).normal(0, 40, size=(16, size))
                return hist, nz.normalize(probe), probe

            res = run_ranks(body, world)
            for r in range(world):
                assert all(np.array_equal(a, b) for h0, h1 in zip(res[0][0], res[r][0]) for a, b in zip(h0, h1))
            hist, normed, probe = res[0]
            # oracle cross-check: world ranks in lockstep with an explicit mean hook
            locals_ = [RunningNorm(size, default_clip_range=5) for _ in range(world)]
            ohist = []
            for step in range(6):
                for r, nz in enumerate(locals_):
                    nz.update(np.clip(_norm_inputs(r, step, size), -200, 200))
                if step % 2 == 1 or step == 4:
                    acc = {}
                    for name in ("local_sum", "local_sumsq", "local_count"):
                        tot = np.zeros_like(getattr(locals_[0], name))
                        for nz in locals_:
                            tot = tot + getattr(nz, name)
                        tot /= world
                        acc[name] = tot
                    for nz in locals_:
                        it = iter([acc["local_sum"], acc["local_sumsq"], acc["local_count"]])
                        nz._mean_over_ranks = lambda x, it=it: next(it).copy()
                        nz.recompute_stats()
                    nz0 = locals_[0]
                    ohist.append([np.array(a, copy=True) for a in (nz0.mean, nz0.std, nz0.total_sum, nz0.total_sumsq, nz0.total_count)])
            for h, oh in zip(hist, ohist):
                for a, b in zip(h, oh):
                    assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), (world, size)
            assert np.array_equal(locals_[0].normalize(probe), normed)
            tag = f"w{world}_d{size}"
            names = ("mean", "std", "total_sum", "total_sumsq", "total_count")
            for i, h in enumerate(hist):
                for nm, a in zip(names, h):
                    payload[f"{tag}_r{i}_{nm}"] = a
            payload[tag + "_n_recompute"] = np.int64(len(hist))
            payload[tag + "_probe"] = probe
            payload[tag + "_normalized"] = normed
    payload["numpy_version"] = np.array(np.__version__)
    payload["std_dtype"] = np.array(str(payload["w1_d27_r0_std"].dtype))
    np.savez_compressed(os.path.join(out, "normalizer.npz"), **payload)
    print("normalizer.npz: std dtype", payload["std_dtype"], "numpy", np.__version__)


def gen_ddpg_update(ref, out):
    """F6: three consecutive _update_network() calls + a polyak update on a seeded agent."""
    import torch

    from oracle import ddpg_update as oupd
    from oracle.her_replay import EpisodeStore, future_probability
    from oracle.running_norm import RunningNorm, update_normalizers
    from rl_arm_under_sparse_reward_amd.synthetic import episode_checksum, make_episodes

    torch.set_num_threads(1)
    env = reference_env()
    env.compute_reward = env.compute_reward.__get__(env)
    args = ref.arguments.Args()
    args.add_demo = False
    args.cuda = False
    args.buffer_size = 64 * 100
    n_eps, dseed, np_seed = 64, 21, 125
    eps = make_episodes(n_eps, seed=dseed, mode="walk")
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(0)
            with quiet():
                agent = ref.ddpg_agent.ddpg_agent(args, env, dict(ENV_PARAMS))
        finally:
            os.chdir(cwd)
    init_actor = {k: v.detach().clone() for k, v in agent.actor_network.state_dict().items()}
    init_critic = {k: v.detach().clone() for k, v in agent.critic_network.state_dict().items()}
    np.random.seed(np_seed)
    agent.buffer.store_episode(eps)
    first_two = [a[:2] for a in eps]
    agent._update_normalizer(first_two)

    grads, losses, batches = [], [], []
    orig_sync = ref.ddpg_agent.sync_grads

    def spy_sync(net):
        grads.append(np.concatenate([p.grad.detach().numpy().ravel() for p in net.parameters()]).astype(np.float32))
        return orig_sync(net)

    ref.ddpg_agent.sync_grads = spy_sync
    orig_backward = torch.Tensor.backward

    def spy_backward(self, *a, **kw):
        losses.append(float(self.detach()))
        return orig_backward(self, *a, **kw)

    torch.Tensor.backward = spy_backward
    orig_sample = agent.buffer.sample

    def spy_sample(bs):
        tr = orig_sample(bs)
        batches.append({k: v.copy() for k, v in tr.items()})
        return tr

    agent.buffer.sample = spy_sample
    try:
        snaps = []
        for _ in range(3):
            agent._update_network()
            snaps.append((ref.utils._get_flat_params(agent.actor_network)[0].copy(),
                          ref.utils._get_flat_params(agent.critic_network)[0].copy()))
        agent._soft_update_target_network(agent.actor_target_network, agent.actor_network)
        agent._soft_update_target_network(agent.critic_target_network, agent.critic_network)
    finally:
        ref.ddpg_agent.sync_grads = orig_sync
        torch.Tensor.backward = orig_backward
    key, pos = np.random.get_state()[1:3]
    tgt_actor = ref.utils._get_flat_params(agent.actor_target_network)[0]
    tgt_critic = ref.utils._get_flat_params(agent.critic_target_network)[0]

    # ---- oracle cross-check of the whole pipeline (store -> norm -> 3x sample+update -> polyak)
    rs = np.random.RandomState(np_seed)
    st = EpisodeStore(100, 27, 3, 4, 64 * 100)
    st.store_episode(eps, rs)
    fp = future_probability("future", args.replay_k)
    on, gn = RunningNorm(27, default_clip_range=5), RunningNorm(3, default_clip_range=5)
    update_normalizers(on, gn, first_two, fp, rs)
    assert np.array_equal(on.mean, agent.o_norm.mean) and np.array_equal(on.std, agent.o_norm.std)
    assert np.array_equal(gn.mean, agent.g_norm.mean) and np.array_equal(gn.std, agent.g_norm.std)
    learner = oupd.DDPGLearner(init_actor, init_critic)
    mb = []
    for i in range(3):
        tr, _ = st.sample(256, fp, rs)
        for kk in tr:
            assert np.array_equal(tr[kk], batches[i][kk]), kk
        x, xn, a, r = oupd.minibatch_tensors(tr, on, gn)
        mb.append((x.numpy(), xn.numpy(), a.numpy(), r.numpy()))
        res = learner.update(x, xn, a, r)
        assert res["actor_loss"] == losses[2 * i] and res["critic_loss"] == losses[2 * i + 1], (res, losses)
        assert np.array_equal(res["actor_grads"], grads[2 * i]) and np.array_equal(res["critic_grads"], grads[2 * i + 1])
        assert np.array_equal(learner.flat("actor"), snaps[i][0]) and np.array_equal(learner.flat("critic"), snaps[i][1])
    learner.soft_update()
    assert np.array_equal(learner.flat("actor_target"), tgt_actor)
    assert np.array_equal(learner.flat("critic_target"), tgt_critic)
    assert np.array_equal(rs.get_state()[1], key) and rs.get_state()[2] == pos

    payload = dict(
        meta=np.array([n_eps, dseed, np_seed, 256, args.replay_k], dtype=np.int64),
        checksum=np.float64(episode_checksum(eps)),
        init_actor=oupd.flatten(list(init_actor.values())), init_critic=oupd.flatten(list(init_critic.values())),
        o_mean=agent.o_norm.mean, o_std=agent.o_norm.std, g_mean=agent.g_norm.mean, g_std=agent.g_norm.std,
        actor_loss=np.array(losses[0::2], dtype=np.float64), critic_loss=np.array(losses[1::2], dtype=np.float64),
        actor_grads_step1=grads[0], critic_grads_step1=grads[1],
        actor_after_step1=snaps[0][0], critic_after_step1=snaps[0][1],
        actor_after_step3=snaps[2][0], critic_after_step3=snaps[2][1],
        actor_target_after_polyak=tgt_actor, critic_target_after_polyak=tgt_critic,
        x_step1=mb[0][0], x_next_step1=mb[0][1], a_step1=mb[0][2], r_step1=mb[0][3],
        key=key.astype(np.uint32), pos=np.int32(pos),
        torch_version=np.array(torch.__version__), numpy_version=np.array(np.__version__),
    )
    np.savez_compressed(os.path.join(out, "ddpg_update.npz"), **payload)
    print("ddpg_update.npz: losses", losses)



def _extract_function(path, name, cls=None):
    """AST node of a top-level function (or a method of class `cls`) of a reference source file."""
    with open(path, encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    for node in tree.body:
        if cls is None and isinstance(node, ast.FunctionDef) and node.name == name:
            return node
        if cls is not None and isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == name:
                    return sub
    raise KeyError(name)


def _reference_agent(ref, n_eps=64, dseed=21, np_seed=125, updates=3):
    """The reference ddpg_agent after store -> _update_normalizer -> `updates` x _update_network."""
    import torch

    from rl_arm_under_sparse_reward_amd.synthetic import make_episodes

    torch.set_num_threads(1)
    env = reference_env()
    env.compute_reward = env.compute_reward.__get__(env)
    args = ref.arguments.Args()
    args.add_demo = False
    args.cuda = False
    args.buffer_size = n_eps * 100
    eps = make_episodes(n_eps, seed=dseed, mode="walk")
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(0)
            with quiet():
                agent = ref.ddpg_agent.ddpg_agent(args, env, dict(ENV_PARAMS))
        finally:
            os.chdir(cwd)
    np.random.seed(np_seed)
    agent.buffer.store_episode(eps)
    agent._update_normalizer([a[:2] for a in eps])
    for _ in range(updates):
        agent._update_network()
    return agent, args


def gen_checkpoint(ref, out):
    """A21 / N2: a checkpoint written by the reference's own save statement (ddpg_agent.py:158-161, lifted out of
    learn() with `ast` and executed unmodified on a reference agent), plus what the reference's reader
    (demo_push.py:15-22 process_inputs + models.actor) computes from it on probe observations."""
    import torch

    agent, args = _reference_agent(ref)
    learn = _extract_function(os.path.join(REF, "ddpg_agent.py"), "learn", cls="ddpg_agent")
    saves = [n for n in ast.walk(learn) if isinstance(n, ast.Expr) and isinstance(n.value, ast.Call)
             and isinstance(n.value.func, ast.Attribute) and n.value.func.attr == "save"
             and getattr(n.value.func.value, "id", "") == "torch"]
    assert len(saves) == 1, "expected exactly one torch.save in ddpg_agent.learn"
    with tempfile.TemporaryDirectory() as tmp:
        agent.model_path = tmp
        agent.savetime = 1
        exec(compile(ast.Module(body=[saves[0]], type_ignores=[]), "ddpg_agent.py:158-161", "exec"),
             {"torch": torch, "self": agent, "str": str})
        name = str(args.seed) + "_" + str(args.add_demo) + "1_model.pt"
        src = os.path.join(tmp, name)
        assert os.path.exists(src), os.listdir(tmp)
        blob = open(src, "rb").read()
    dst = os.path.join(out, "ref_checkpoint_model.pt")
    with open(dst, "wb") as f:
        f.write(blob)
    # the reference's reader: demo_push.py:15-22 (process_inputs, executed unmodified) + models.actor
    proc = _extract_function(os.path.join(REF, "demo_push.py"), "process_inputs")
    ns = {"np": np, "torch": torch}
    exec(compile(ast.Module(body=[proc], type_ignores=[]), "demo_push.py:15-22", "exec"), ns)
    o_mean, o_std, g_mean, g_std, model = torch.load(dst, map_location=lambda storage, loc: storage, weights_only=False)
    net = ref.models.actor(dict(ENV_PARAMS))
    net.load_state_dict(model)
    net.eval()
    rs = np.random.RandomState(77)
    probe_o = rs.uniform(-1, 1, (33, 27))
    probe_o[3] *= 400.0                                   # beyond clip_obs
    probe_g = rs.uniform(-0.2, 0.7, (33, 3))
    acts, xs = [], []
    with torch.no_grad():
        for o, g in zip(probe_o, probe_g):
            x = ns["process_inputs"](o, g, o_mean, o_std, g_mean, g_std, args)
            xs.append(x.numpy())
            acts.append(net(x).numpy().squeeze())
    meta = {k: (tuple(v.shape), str(v.dtype)) for k, v in model.items()}
    np.savez_compressed(os.path.join(out, "ref_checkpoint_probe.npz"), probe_obs=probe_o, probe_g=probe_g,
                        inputs=np.array(xs), actions=np.array(acts), o_mean=o_mean, o_std=o_std, g_mean=g_mean,
                        g_std=g_std, keys=np.array(list(model.keys())),
                        shapes=np.array([str(meta[k][0]) for k in model]),
                        clip_obs=np.float64(args.clip_obs), clip_range=np.float64(args.clip_range),
                        torch_version=np.array(torch.__version__))
    print("ref_checkpoint_model.pt:", len(blob), "bytes; keys", list(model.keys()))


def gen_reference_demo(out, demo_num=6):
    """A12 / N2: a demo file written by the reference's own generator: get_push_demo (get_demo_data_push.py:24-94,
    executed unmodified; its module cannot be imported because it pulls in gym + pybullet at the top) driving a
    stand-in GoalEnv.  The episode contents come from the stand-in; the SCHEMA -- keys, nesting, squeeze, dtypes,
    the object array of info dicts -- is whatever the reference writer produces."""
    import math

    from rl_arm_under_sparse_reward_amd.synthetic import PointMassGoalEnv

    fn = _extract_function(os.path.join(REF, "get_demo_data_push.py"), "get_push_demo")
    ns = {"np": np, "math": math, "demo_num": demo_num, "print": lambda *a, **k: None}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "get_demo_data_push.py:24-94", "exec"), ns)
    env = PointMassGoalEnv(seed=4)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            ns["get_push_demo"](env, env.env_params)
        finally:
            os.chdir(cwd)
        files = os.listdir(tmp)
        assert files == [f"bmirobot_{demo_num}_push_demo.npz"], files
        blob = open(os.path.join(tmp, files[0]), "rb").read()
    dst = os.path.join(out, f"ref_written_{demo_num}_push_demo.npz")
    with open(dst, "wb") as f:
        f.write(blob)
    d = np.load(dst, allow_pickle=True)
    print(os.path.basename(dst), {k: (d[k].shape, str(d[k].dtype)) for k in d.files})


def reference_env_full(reward_type="sparse", distance_threshold=0.05):
    """Dummy env carrying the reference's compute_reward AND _is_success bodies (bmirobot_env_push_F.py:84-90,243-245)."""
    path = os.path.join(REF, "bmirobot_env", "bmirobot_env_push_F.py")
    gd = _extract_function(path, "goal_distance")
    cr = _extract_function(path, "compute_reward", cls="bmirobotGymEnv")
    su = _extract_function(path, "_is_success", cls="bmirobotGymEnv")
    ns = {"np": np}
    exec(compile(ast.Module(body=[gd, cr, su], type_ignores=[]), path, "exec"), ns)

    class _Env:
        pass

    e = _Env()
    e.reward_type, e.distance_threshold = reward_type, distance_threshold
    e.compute_reward = ns["compute_reward"].__get__(e)
    e._is_success = ns["_is_success"].__get__(e)
    return e


def gen_dense_reward(out):
    """N4: the dense branch of compute_reward (-d, float64) and _is_success (d < thr, float32) on the adversarial pairs
    of F3 plus random ones, for two thresholds; outputs of the reference functions themselves."""
    adv = np.load(os.path.join(out, "reward_adversarial.npz"))
    rs = np.random.RandomState(6)
    ag = np.concatenate([adv["ag"], rs.uniform(-1, 1, (1024, 3))])
    g = np.concatenate([adv["g"], rs.uniform(-1, 1, (1024, 3))])
    payload = {"ag": ag, "g": g}
    for thr in (0.05, 0.1):
        dense = reference_env_full("dense", thr)
        sparse = reference_env_full("sparse", thr)
        rd = dense.compute_reward(ag, g, None)
        sp = sparse.compute_reward(ag, g, None)
        ok = sparse._is_success(ag, g)
        assert rd.dtype == np.float64 and sp.dtype == np.float32 and ok.dtype == np.float32
        tag = f"thr{thr}"
        payload[tag + "_dense"] = rd
        payload[tag + "_sparse_bits"] = sp.view(np.uint32)
        payload[tag + "_success"] = ok
    # leading-dims vectorisation (her.py:38 passes [B, 3]; rollouts pass [3])
    payload["stack_ag"] = ag[:60].reshape(5, 4, 3, 3)[..., 0, :].copy()
    payload["stack_g"] = g[:60].reshape(5, 4, 3, 3)[..., 0, :].copy()
    payload["stack_dense"] = reference_env_full("dense").compute_reward(payload["stack_ag"], payload["stack_g"], None)
    np.savez_compressed(os.path.join(out, "reward_dense_success.npz"), **payload)
    print("reward_dense_success.npz:", ag.shape[0], "pairs; successes at 0.05:", int(payload["thr0.05_success"].sum()))


ROLLOUT_CFG = dict(env_seed=11, np_seed=5, torch_seed=0, n_epochs=2, n_cycles=2, n_batches=3, n_test_rollouts=12,
                   noise_eps=0.05, random_eps=0.3, buffer_episodes=64, distance_threshold=0.25)


def gen_rollout(ref, out):
    """N1 / N3: the reference's own learn() (rollout loop, exploration draws, store, normalizer, updates, polyak,
    _eval_agent, checkpoint) on the package's stand-in GoalEnv; every episode batch it stores, its success rates, the
    RNG state it ends in and its final parameters.  Cross-checked bit for bit against oracle/rollout.py."""
    import torch

    from oracle.rollout import OracleAgent
    from rl_arm_under_sparse_reward_amd.synthetic import PointMassGoalEnv

    torch.set_num_threads(1)
    c = ROLLOUT_CFG
    env = PointMassGoalEnv(seed=c["env_seed"], max_timesteps=100, distance_threshold=c["distance_threshold"])
    env_params = env.env_params
    args = ref.arguments.Args()
    args.n_epochs, args.n_cycles, args.n_batches = c["n_epochs"], c["n_cycles"], c["n_batches"]
    args.n_test_rollouts, args.noise_eps, args.random_eps = c["n_test_rollouts"], c["noise_eps"], c["random_eps"]
    args.buffer_size, args.add_demo, args.cuda = c["buffer_episodes"] * 100, False, False
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            torch.manual_seed(c["torch_seed"])
            with quiet():
                agent = ref.ddpg_agent.ddpg_agent(args, env, dict(env_params))
            init_actor = {k: v.detach().clone() for k, v in agent.actor_network.state_dict().items()}
            init_critic = {k: v.detach().clone() for k, v in agent.critic_network.state_dict().items()}
            stored = []
            orig_store = agent.buffer.store_episode

            def spy(batch):
                stored.append([np.array(a, copy=True) for a in batch])
                return orig_store(batch)

            agent.buffer.store_episode = spy
            np.random.seed(c["np_seed"])
            with quiet():
                agent.learn()
            saved = sorted(os.listdir(agent.model_path))
        finally:
            os.chdir(cwd)
    key, pos = np.random.get_state()[1:3]
    assert len(stored) == c["n_epochs"] * c["n_cycles"] and len(saved) == c["n_epochs"]
    assert stored[0][3].dtype == np.float32      # the reference stores float32 actions (in-place `action +=`, :177)
    # ---- oracle cross-check: same seeds, same env, bit for bit
    env2 = PointMassGoalEnv(seed=c["env_seed"], max_timesteps=100, distance_threshold=c["distance_threshold"])
    np.random.seed(c["np_seed"])
    oa = OracleAgent(env2, env_params, init_actor, init_critic, buffer_size=args.buffer_size, batch_size=args.batch_size,
                     replay_k=args.replay_k, n_batches=args.n_batches, num_rollouts=args.num_rollouts_per_mpi,
                     n_test_rollouts=args.n_test_rollouts, noise_eps=args.noise_eps, random_eps=args.random_eps)
    oa.learn_epochs(c["n_epochs"], c["n_cycles"])
    for a, b in zip(stored, oa.episodes):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and np.array_equal(x.view(np.uint8), y.view(np.uint8))
    assert [float(r) for r in agent.success_rates] == [float(r) for r in oa.success_rates]
    assert np.array_equal(np.random.get_state()[1], key) and np.random.get_state()[2] == pos
    actor_final = ref.utils._get_flat_params(agent.actor_network)[0]
    assert np.array_equal(actor_final, oa.learner.flat("actor"))
    from oracle import ddpg_update as oupd
    payload = dict(cfg=np.array(sorted(c.items()), dtype=object).astype(str), init_actor=oupd.flatten(list(init_actor.values())),
                   init_critic=oupd.flatten(list(init_critic.values())), success_rates=np.array(agent.success_rates, np.float64),
                   actor_final=actor_final, critic_final=ref.utils._get_flat_params(agent.critic_network)[0],
                   o_mean=agent.o_norm.mean, o_std=agent.o_norm.std, g_mean=agent.g_norm.mean, g_std=agent.g_norm.std,
                   key=key.astype(np.uint32), pos=np.int32(pos), checkpoints=np.array(saved))
    for i, b in enumerate(stored):
        for nm, a in zip(("obs", "ag", "g", "actions"), b):
            payload[f"cycle{i}_{nm}"] = a
    np.savez_compressed(os.path.join(out, "rollout.npz"), **payload)
    print("rollout.npz:", len(stored), "stored batches; success rates", [float(r) for r in agent.success_rates], saved)

def gen_demo(out):
    from rl_arm_under_sparse_reward_amd.synthetic import write_demo_npz

    write_demo_npz(os.path.join(out, "bmirobot_8_push_demo.npz"), n_episodes=8, seed=7)
    print("bmirobot_8_push_demo.npz written (synthetic; the real 1000-episode files are absent from the mount)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    ref = load_reference()
    todo = a.only.split(",") if a.only else ["rng", "her", "reward", "storage", "norm", "ddpg", "demo", "ckpt", "refdemo",
                                             "dense", "rollout"]
    if "rng" in todo:
        gen_rng_kat(ref, a.out)
    if "her" in todo:
        gen_her_samples(ref, a.out)
    if "reward" in todo:
        gen_reward_adversarial(a.out)
    if "storage" in todo:
        gen_storage_idx(ref, a.out)
    if "norm" in todo:
        gen_normalizer(ref, a.out)
    if "ddpg" in todo:
        gen_ddpg_update(ref, a.out)
    if "demo" in todo:
        gen_demo(a.out)
    if "ckpt" in todo:
        gen_checkpoint(ref, a.out)
    if "refdemo" in todo:
        gen_reference_demo(a.out)
    if "dense" in todo:
        gen_dense_reward(a.out)
    if "rollout" in todo:
        gen_rollout(ref, a.out)


if __name__ == "__main__":
    main()
