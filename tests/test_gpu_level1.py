"""INTEGRATION.md section 1 ("import swap of the data path"), end to end.

The reference's `ddpg_agent._update_network` (ddpg_agent.py:225-277) keeps its torch learner -- autograd, torch.optim.Adam -- and
only the DATA PATH objects are the mirror's: `replay_buffer.sample()` -> `_preproc_og` -> `normalizer.normalize()` -> torch ->
`sync_grads` right behind each backward, `sync_networks` at construction (ddpg_agent.py:27-28).  Run in the reference's call
order for a whole cycle (ddpg_agent.py:143-150: store_episode, _update_normalizer, 40 updates, polyak), it must be
BITWISE equal to the same learner fed by the CPU oracle: every minibatch row, every loss, every parameter, and the random
stream to the last word.  (The learner is oracle.ddpg_update.DDPGLearner on both sides -- the torch arithmetic is identical by
construction; what the test pins is that the device data path hands it identical inputs at every step.)"""
import numpy as np
import pytest
import torch

from conftest import bits
from gpu_common import ENV_PARAMS, ctx, fresh_rng, state_equal
from oracle import ddpg_update as oupd
from oracle.her_replay import EpisodeStore, future_probability
from oracle.running_norm import RunningNorm, preproc_og, update_normalizers
from rl_arm_under_sparse_reward_amd.her import her_sampler
from rl_arm_under_sparse_reward_amd.normalizer import normalizer
from rl_arm_under_sparse_reward_amd.replay_buffer import replay_buffer
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes
from rl_arm_under_sparse_reward_amd.utils import sync_grads, sync_networks

pytestmark = pytest.mark.gpu


class _Net:
    """What utils.sync_grads / sync_networks take: an object with named_parameters() (utils.py:18-27 order)."""

    def __init__(self, params):
        self._p = params

    def named_parameters(self):
        return list(self._p.items())


class _Level1Learner(oupd.DDPGLearner):
    """The oracle learner with the reference's exchange calls routed through the MIRROR's utils (ddpg_agent.py:271, :276)."""

    def _sync_grads(self, params):
        sync_grads(_Net(params))


def _reference_update(agent_o_norm, agent_g_norm, buffer, learner, batch, clip_obs=200):
    """ddpg_agent.py:227-248 statement for statement, on the mirror's objects; returns the four tensors it feeds the networks."""
    transitions = buffer.sample(batch)                                                   # :227
    o, o_next, g = transitions['obs'], transitions['obs_next'], transitions['g']         # :229
    transitions['obs'], transitions['g'] = np.clip(o, -clip_obs, clip_obs), np.clip(g, -clip_obs, clip_obs)             # :230
    transitions['obs_next'], transitions['g_next'] = np.clip(o_next, -clip_obs, clip_obs), np.clip(g, -clip_obs, clip_obs)   # :231
    obs_norm = agent_o_norm.normalize(transitions['obs'])                                 # :233
    g_norm = agent_g_norm.normalize(transitions['g'])
    inputs_norm = np.concatenate([obs_norm, g_norm], axis=1)                              # :235
    obs_next_norm = agent_o_norm.normalize(transitions['obs_next'])
    g_next_norm = agent_g_norm.normalize(transitions['g_next'])
    inputs_next_norm = np.concatenate([obs_next_norm, g_next_norm], axis=1)               # :238
    x = torch.tensor(inputs_norm, dtype=torch.float32)                                    # :240-243
    x_next = torch.tensor(inputs_next_norm, dtype=torch.float32)
    actions = torch.tensor(transitions['actions'], dtype=torch.float32)
    r = torch.tensor(transitions['r'], dtype=torch.float32)
    return learner.update(x, x_next, actions, r), (x, x_next, actions, r)


@pytest.mark.parametrize("batch,k,n_cycles", [(256, 4, 2), (64, 8, 1)])
def test_reference_learner_on_the_mirror_data_path_is_bitwise_the_oracle_run(batch, k, n_cycles):
    torch.set_num_threads(4)
    seed, n_eps, n_batches = 125, 40, 40
    a0 = oupd.init_actor(27, 3, 4, seed=1)
    c0 = oupd.init_critic(27, 3, 4, seed=2)
    fp = future_probability("future", k)
    first = make_episodes(n_eps, seed=3, mode="walk")
    # ---- all-oracle run
    rs = np.random.RandomState(seed)
    store = EpisodeStore(100, 27, 3, 4, n_eps * 100 + 300)
    on, gn = RunningNorm(27, default_clip_range=5), RunningNorm(3, default_clip_range=5)
    want = oupd.DDPGLearner(a0, c0)
    store.store_episode(first, rs)
    want_log, want_x, want_stats = [], [], []
    for c in range(n_cycles):
        eps = make_episodes(2, seed=50 + c, mode="walk")
        store.store_episode(eps, rs)                                   # ddpg_agent.py:143
        update_normalizers(on, gn, eps, fp, rs)                        # :144
        want_stats.append((on.mean.copy(), on.std.copy(), gn.mean.copy(), gn.std.copy()))
        for _ in range(n_batches):                                     # :145-147
            tr, _ = store.sample(batch, fp, rs)
            mb = oupd.minibatch_tensors(tr, on, gn)
            res = want.update(*mb)
            want_log.append((res["actor_loss"], res["critic_loss"]))
            want_x.append(mb)
        want.soft_update()                                             # :149-150
    # ---- the reference's learner on the mirror's data path
    rng = fresh_rng(seed)
    her = her_sampler("future", k, None, rng=rng)
    buf = replay_buffer(dict(ENV_PARAMS), n_eps * 100 + 300, her.sample_her_transitions, rng=rng, ctx=ctx())
    o_norm = normalizer(size=27, default_clip_range=5, ctx=ctx())
    g_norm = normalizer(size=3, default_clip_range=5, ctx=ctx())
    got = _Level1Learner(a0, c0)
    sync_networks(_Net(got.actor))                                     # ddpg_agent.py:27-28
    sync_networks(_Net(got.critic))
    buf.store_episode(first)
    i = 0
    for c in range(n_cycles):
        mb_obs, mb_ag, mb_g, mb_actions = make_episodes(2, seed=50 + c, mode="walk")
        buf.store_episode([mb_obs, mb_ag, mb_g, mb_actions])           # :143
        # _update_normalizer (:187-212) on the mirror's sampler and normalizers
        buffer_temp = {'obs': mb_obs, 'ag': mb_ag, 'g': mb_g, 'actions': mb_actions,
                       'obs_next': mb_obs[:, 1:, :], 'ag_next': mb_ag[:, 1:, :]}
        transitions = her.sample_her_transitions(buffer_temp, mb_actions.shape[1])
        obs, g = preproc_og(transitions['obs'], transitions['g'], 200)
        o_norm.update(obs)
        g_norm.update(g)
        o_norm.recompute_stats()
        g_norm.recompute_stats()
        for a, b in zip((o_norm.mean, o_norm.std, g_norm.mean, g_norm.std), want_stats[c]):
            assert a.dtype == b.dtype and np.array_equal(bits(a), bits(b)), ("normalizer statistics", c)
        for _ in range(n_batches):
            res, mb = _reference_update(o_norm, g_norm, buf, got, batch)
            for a, b in zip(mb, want_x[i]):
                assert np.array_equal(bits(a.numpy()), bits(b.numpy())), ("minibatch", i)
            assert (res["actor_loss"], res["critic_loss"]) == want_log[i], (i, res["actor_loss"], want_log[i])
            i += 1
        got.soft_update()
    for which in ("actor", "critic", "actor_target", "critic_target"):
        assert np.array_equal(bits(got.flat(which)), bits(want.flat(which))), which
    assert state_equal(rng, *rs.get_state()[1:3])
    assert buf.current_size == store.current_size and buf.n_transitions_stored == store.n_transitions_stored


def test_torch_learner_on_the_gpu_fed_by_the_device_output_sampler():
    """INTEGRATION.md section 1 with `args.cuda` (ddpg_agent.py:244-248): the reference's torch learner runs ON THE GPU and its
    four input tensors come straight out of `replay_buffer.sample_device()` (hp_buffer_sample_dev: gather + relabel + reward + clip +
    normalise -> float32 in device memory) -- nothing of a minibatch crosses PCIe.  Every minibatch must be BITWISE what the
    all-oracle CPU run feeds its learner (same draws, same float64 arithmetic, same float32 rounding), the random stream must end
    on the same word; the losses follow within the north-star 1e-5 on the first update and a drift bound after (the GPU's
    matrix products sum in another order than the CPU's)."""
    torch.set_num_threads(4)
    seed, n_eps, n_batches, batch, k = 125, 40, 40, 256, 4
    a0 = oupd.init_actor(27, 3, 4, seed=1)
    c0 = oupd.init_critic(27, 3, 4, seed=2)
    fp = future_probability("future", k)
    first = make_episodes(n_eps, seed=3, mode="walk")
    eps = make_episodes(2, seed=50, mode="walk")
    # ---- all-oracle run (CPU)
    rs = np.random.RandomState(seed)
    store = EpisodeStore(100, 27, 3, 4, n_eps * 100 + 300)
    on, gn = RunningNorm(27, default_clip_range=5), RunningNorm(3, default_clip_range=5)
    want = oupd.DDPGLearner(a0, c0)
    store.store_episode(first, rs)
    store.store_episode(eps, rs)
    update_normalizers(on, gn, eps, fp, rs)
    want_x, want_log = [], []
    for _ in range(n_batches):
        tr, _ = store.sample(batch, fp, rs)
        mb = oupd.minibatch_tensors(tr, on, gn)
        res = want.update(*mb)
        want_x.append(mb)
        want_log.append((res["actor_loss"], res["critic_loss"]))
    # ---- torch learner on cuda:0, mirror data path with device outputs
    dev = torch.device("cuda", ctx().device_id)
    rng = fresh_rng(seed)
    her = her_sampler("future", k, None, rng=rng)
    buf = replay_buffer(dict(ENV_PARAMS), n_eps * 100 + 300, her.sample_her_transitions, rng=rng, ctx=ctx())
    o_norm = normalizer(size=27, default_clip_range=5, ctx=ctx())
    g_norm = normalizer(size=3, default_clip_range=5, ctx=ctx())
    got = oupd.DDPGLearner({kk: v.to(dev) for kk, v in a0.items()}, {kk: v.to(dev) for kk, v in c0.items()})
    buf.store_episode(first)
    buf.store_episode(eps)
    mb_obs, mb_ag, mb_g, mb_actions = eps
    transitions = her.sample_her_transitions({'obs': mb_obs, 'ag': mb_ag, 'g': mb_g, 'actions': mb_actions,
                                              'obs_next': mb_obs[:, 1:, :], 'ag_next': mb_ag[:, 1:, :]}, mb_actions.shape[1])
    obs, g = preproc_og(transitions['obs'], transitions['g'], 200)
    o_norm.update(obs); g_norm.update(g)
    o_norm.recompute_stats(); g_norm.recompute_stats()
    assert np.array_equal(bits(o_norm.mean), bits(on.mean)) and np.array_equal(bits(g_norm.std), bits(gn.std))
    for i in range(n_batches):
        mb = buf.sample_device(batch, o_norm, g_norm, clip_obs=200)                      # ddpg_agent.py:227-243 in one kernel
        assert all(t.is_cuda and t.dtype == torch.float32 for t in mb.values())
        for key, ref in zip(("x", "x_next", "actions", "r"), want_x[i]):
            assert np.array_equal(bits(mb[key].cpu().numpy()), bits(ref.numpy())), ("minibatch", i, key)
        res = got.update(mb["x"], mb["x_next"], mb["actions"], mb["r"])                  # :250-277 on the GPU
        for j, name in enumerate(("actor_loss", "critic_loss")):
            tol = 1e-5 if i == 0 else 1e-5 * 1.3 ** min(i, 20)
            assert abs(res[name] - want_log[i][j]) <= tol * max(abs(want_log[i][j]), 1e-2), (i, name, res[name], want_log[i][j])
    assert state_equal(rng, *rs.get_state()[1:3])


def test_sample_device_leaves_the_fused_learner_on_its_graphs():
    """sample_device() used to rebind the shared Context to torch's stream -- hipStreamLegacy for torch's
    default -- and leave it there, after which hp_agent_train_cycle refused graphs for good.  Now the sampler borrows torch's
    stream for its own call only (hp_ctx_borrow_stream / hp_ctx_return_stream): a fused learner on the same context, interleaved
    with sample_device() calls consumed by torch, keeps replaying its cycle graph and ends bit-identical to the learner that was
    never interleaved (the sampler's draws are rewound so that both see the same random stream)."""
    import ctypes as C
    from rl_arm_under_sparse_reward_amd import _lib
    from rl_arm_under_sparse_reward_amd.arguments import Args
    from rl_arm_under_sparse_reward_amd.ddpg_agent import NET_ACTOR, NET_CRITIC, ddpg_agent

    def run(interleave):
        torch.manual_seed(0)
        rng = fresh_rng(31)
        agent = ddpg_agent(Args(batch_size=256, buffer_size=32 * 100), None, dict(ENV_PARAMS), ctx=ctx(), rng=rng)
        agent.buffer.store_episode(make_episodes(20, seed=5, mode="walk"))
        sums = []
        for c in range(4):
            agent.train_cycle(make_episodes(2, seed=90 + c, mode="walk"), 12)
            if interleave:
                state = rng.get_state()
                mb = agent.buffer.sample_device(256, agent.o_norm, agent.g_norm, clip_obs=200)
                sums.append(float((mb["x"].double().sum() + mb["r"].double().sum()).item()))     # consumed on torch's stream
                rng.set_state(state)
        mode, stream = C.c_int32(), C.c_void_p()
        _lib.check(agent.lib.hp_agent_cycle_mode(agent.h, C.byref(mode)))
        _lib.check(agent.lib.hp_ctx_get_stream(agent.ctx.h, C.byref(stream)))
        return agent._get_flat(NET_ACTOR), agent._get_flat(NET_CRITIC), agent.last_losses(12), mode.value, stream.value or 0, sums

    plain = run(False)
    mixed = run(True)
    assert plain[3] == 1 and mixed[3] == 1                      # the cycle replays as a hipGraph in both
    assert mixed[4] == plain[4] and mixed[4] > 2                # the context is still on its own stream (not hipStreamLegacy = 1)
    for a, b in zip(plain[:3], mixed[:3]):
        assert np.array_equal(bits(np.asarray(a)), bits(np.asarray(b)))
    assert len(mixed[5]) == 4 and all(np.isfinite(v) for v in mixed[5])


def test_borrow_and_return_stream_contract():
    """hp_ctx_borrow_stream / hp_ctx_return_stream (ABI 4): one borrow at a time, a return needs a borrow, borrowing the stream the
    context is on already is a no-op, and after the return the context is on the stream it was on before -- with the work it
    enqueues next ordered behind what ran on the borrowed stream (a store behind a sample behind a store, read back)."""
    import ctypes as C
    from rl_arm_under_sparse_reward_amd import _lib
    from gpu_common import DeviceEpisodeBuffer

    c = ctx()
    lib = c.lib
    mine = C.c_void_p()
    _lib.check(lib.hp_ctx_get_stream(c.h, C.byref(mine)))
    assert (mine.value or 0) > 2                                    # the context's own stream
    assert lib.hp_ctx_return_stream(c.h) != 0                       # nothing borrowed
    side = torch.cuda.Stream()
    _lib.check(lib.hp_ctx_borrow_stream(c.h, C.c_void_p(side.cuda_stream)))
    now = C.c_void_p()
    _lib.check(lib.hp_ctx_get_stream(c.h, C.byref(now)))
    assert now.value == side.cuda_stream
    assert lib.hp_ctx_borrow_stream(c.h, C.c_void_p(side.cuda_stream)) != 0      # one at a time
    _lib.check(lib.hp_ctx_return_stream(c.h))
    _lib.check(lib.hp_ctx_get_stream(c.h, C.byref(now)))
    assert now.value == mine.value
    _lib.check(lib.hp_ctx_borrow_stream(c.h, mine))                 # the stream it is on: nothing to order, nothing left foreign
    _lib.check(lib.hp_ctx_return_stream(c.h))
    # ordering across a borrow: store (own stream) -> sample on a side stream -> store again (own stream) -> read back
    rng = fresh_rng(3)
    buf = DeviceEpisodeBuffer(4, 100, 27, 3, 4)
    first, second = make_episodes(4, seed=1, mode="walk"), make_episodes(4, seed=2, mode="walk")
    o_dev, g_dev = normalizer(27, default_clip_range=5, ctx=c), normalizer(3, default_clip_range=5, ctx=c)
    buf.store(rng, first)
    with torch.cuda.stream(side):
        from rl_arm_under_sparse_reward_amd.her import squared_threshold
        a = buf.sample_device(rng, o_dev, g_dev, 4096, 0.8, squared_threshold(0.05), 200)        # reads `first` on the side stream
    buf.store(rng, second)                                                       # a full buffer: random slots overwritten, behind the sample
    side.synchronize()
    c.synchronize()
    rs = np.random.RandomState(3)
    from oracle.her_replay import EpisodeStore
    st = EpisodeStore(100, 27, 3, 4, 400)
    st.store_episode(first, rs)
    ref, _ = st.sample(4096, 0.8, rs)
    from oracle.running_norm import RunningNorm
    on, gn = RunningNorm(27, default_clip_range=5), RunningNorm(3, default_clip_range=5)
    from oracle.ddpg_update import minibatch_tensors
    x, _, _, _ = minibatch_tensors(ref, on, gn, 200)
    assert np.array_equal(bits(a["x"].cpu().numpy()), bits(x.numpy()))          # the sample saw `first`, whole
    st.store_episode(second, rs)
    assert np.array_equal(buf.read("obs", 0, 4), st.buffers["obs"][:4])        # and the second store landed after it
    assert state_equal(rng, *rs.get_state()[1:3])
