"""HIP replay buffer + HER sampler vs the oracle and the reference-generated fixtures.
Bit-exact: sampled rows, relabelled goals, reward bit patterns, storage slots, RNG state."""
import numpy as np
import pytest

from conftest import bits, load_golden
from gpu_common import ENV_PARAMS, DeviceEpisodeBuffer, fresh_rng, state_equal
from oracle.her_replay import EpisodeStore, compute_reward, future_probability
from rl_arm_under_sparse_reward_amd.her import her_sampler, squared_threshold
from rl_arm_under_sparse_reward_amd.replay_buffer import replay_buffer
from rl_arm_under_sparse_reward_amd.synthetic import episode_checksum, make_episodes

pytestmark = pytest.mark.gpu
KEYS = ("obs", "ag", "g", "actions", "obs_next", "ag_next", "r")


def test_her_sample_golden_bitwise():
    g = load_golden("her_sample.npz")
    for tag in g["cases"]:
        tag = str(tag)
        n, B, k, seed, dseed = (int(x) for x in g[tag + "_meta"])
        eps = make_episodes(n, seed=dseed, mode=str(g[tag + "_mode"]))
        assert episode_checksum(eps) == float(g[tag + "_checksum"])
        dev = fresh_rng(seed)
        sampler = her_sampler("future", k, rng=dev)
        buf = replay_buffer(ENV_PARAMS, n * 100, sampler.sample_her_transitions, rng=dev)
        buf.store_episode(eps)
        assert buf.current_size == n and buf.n_transitions_stored == 100 * n
        tr = buf.sample(B)
        assert set(tr) == set(KEYS)
        for key in KEYS:
            ref = g[f"{tag}_{key}"]
            assert tr[key].dtype == ref.dtype and tr[key].shape == ref.shape, (tag, key)
            assert np.array_equal(bits(tr[key]), bits(ref)), (tag, key)
        assert state_equal(dev, g[tag + "_key"], g[tag + "_pos"]), tag


@pytest.mark.parametrize("n,B,k,mode", [(5000, 256, 4, "iid"), (5000, 4096, 8, "walk"), (37, 1000, 4, "walk")])
def test_sample_matches_oracle_at_baseline_sizes(n, B, k, mode):
    """BASELINE.json configs: buffer 5e5 (5000 episodes), B in {256, 4096}, replay_k in {4, 8}."""
    eps = make_episodes(n, seed=1, mode=mode)
    fp = future_probability("future", k)
    st = EpisodeStore(100, 27, 3, 4, n * 100)
    rs = np.random.RandomState(125)
    st.store_episode(eps, rs)
    dev = fresh_rng(125)
    buf = DeviceEpisodeBuffer(n, 100, 27, 3, 4)
    buf.store(dev, eps)
    for _ in range(3):
        ref, ridx = st.sample(B, fp, rs)
        tr, idx = buf.sample(dev, B, fp, squared_threshold(0.05), with_indices=True)
        for key in KEYS:
            assert np.array_equal(bits(tr[key]), bits(ref[key])), key
        assert np.array_equal(idx["e"], ridx["e"]) and np.array_equal(idx["t"], ridx["t"])
        assert np.array_equal(idx["her"], ridx["her"]) and np.array_equal(idx["future_t"], ridx["future_t"])
    assert state_equal(dev, *rs.get_state()[1:3])
    # size-independent properties (hold at any size): relabelled goals are achieved goals of the same
    # episode at a strictly later step, and a goal relabelled to t+1 always succeeds (-0.0)
    e, t, fut, her = idx["e"], idx["t"], idx["future_t"], idx["her"]
    assert np.array_equal(tr["g"][her], eps[1][e[her], fut[her]])
    assert np.array_equal(tr["g"][~her], eps[2][e[~her], t[~her]])
    nxt = her & (fut == t + 1)
    assert np.all(tr["r"][nxt].view(np.uint32) == 0x80000000)
    assert set(np.unique(tr["r"].view(np.uint32))) <= {0x80000000, 0xBF800000}


def test_reward_bits_on_adversarial_pairs():
    """F3: distances within a few ulps of the 0.05 radius, d == 0, denormal differences."""
    g = load_golden("reward_adversarial.npz")
    ag, goal = g["ag"], g["g"]
    M = ag.shape[0]
    obs = np.zeros((M, 2, 1))
    agm = np.zeros((M, 2, 3)); agm[:, 1, :] = ag
    dev = fresh_rng(11)
    buf = DeviceEpisodeBuffer(M, 1, 1, 3, 1)
    buf.store(dev, [obs, agm, goal[:, None, :], np.zeros((M, 1, 1))])
    seen = np.zeros(M, bool)
    for _ in range(6):
        tr, idx = buf.sample(dev, 8192, 0.0, squared_threshold(0.05), with_indices=True)
        assert np.array_equal(tr["r"][:, 0].view(np.uint32), g["r_bits"][idx["e"]])
        seen[idx["e"]] = True
    assert seen.mean() > 0.99


def test_storage_slots_golden_and_contents():
    g = load_golden("storage_idx.npz")
    for tag in g["cases"]:
        tag = str(tag)
        size, seed = int(g[tag + "_size"]), int(g[tag + "_seed"])
        dev = fresh_rng(seed)
        rs = np.random.RandomState(seed)
        buf = DeviceEpisodeBuffer(size, 3, 2, 1, 1)
        st = EpisodeStore(3, 2, 1, 1, size * 3)
        slots, sizes = [], []
        data_rs = np.random.RandomState(1)
        for inc in g[tag + "_incs"]:
            inc = int(inc)
            eps = [data_rs.normal(size=(inc, 4, 2)), data_rs.normal(size=(inc, 4, 1)),
                   data_rs.normal(size=(inc, 3, 1)), data_rs.normal(size=(inc, 3, 1))]
            buf.store(dev, eps)
            st.store_episode(eps, rs)
            slots.append(buf.last_slots(inc))
            sizes.append(buf.info()[1])
        assert np.array_equal(np.concatenate(slots), g[tag + "_slots"]), tag
        assert np.array_equal(sizes, g[tag + "_current_size"]), tag
        assert state_equal(dev, g[tag + "_key"], g[tag + "_pos"]), tag
        cur = sizes[-1]
        for key in ("obs", "ag", "g", "actions"):   # repeated slots: the LAST episode wins, like numpy
            assert np.array_equal(buf.read(key, 0, cur), st.buffers[key][:cur]), (tag, key)


def test_empty_and_invalid_calls_raise_like_the_reference():
    dev = fresh_rng(0)
    sampler = her_sampler("future", 4, rng=dev)
    buf = replay_buffer(ENV_PARAMS, 1000, sampler.sample_her_transitions, rng=dev)
    with pytest.raises(ValueError, match="high <= 0"):
        buf.sample(4)                                   # np.random.randint(0, 0, 4) in the reference
    with pytest.raises(ValueError, match="high <= 0"):
        buf.store_episode(make_episodes(11, seed=1))    # demo larger than an empty buffer
    with pytest.raises(ValueError):
        buf.store_episode([np.zeros((1, 100, 27)), np.zeros((1, 101, 3)), np.zeros((1, 100, 3)), np.zeros((1, 100, 4))])
    with pytest.raises(TypeError):
        replay_buffer(ENV_PARAMS, 1000, lambda b, n: b)  # no host sample functions
    with pytest.raises(NotImplementedError):
        her_sampler("future", 4, reward_type="shaped")   # only compute_reward's two branches run on the device


def test_reward_callables_are_probed_not_trusted():
    """her.py:11,38: the reference CALLS whatever reward_func it is given.  The device computes the goal-distance reward
    described by (distance_threshold, reward_type); a callable that is anything else must raise at construction, never be
    silently replaced by sparse / 0.05."""
    from rl_arm_under_sparse_reward_amd.synthetic import PointMassGoalEnv
    dev = fresh_rng(3)

    def ref_sparse(thr):
        return lambda ag, g, info: -(np.linalg.norm(ag - g, axis=-1) > thr).astype(np.float32)

    # plain callables with no bound env: accepted when they ARE the described reward ...
    assert her_sampler("future", 4, ref_sparse(0.05), rng=dev).distance_threshold == 0.05
    s = her_sampler("future", 4, ref_sparse(0.07), distance_threshold=0.07, rng=dev)
    assert s.sq_threshold == pytest.approx(0.07 ** 2, rel=1e-12)
    her_sampler("future", 4, lambda ag, g, info: -np.linalg.norm(ag - g, axis=-1), reward_type="dense", rng=dev)
    # ... refused when they are not: another threshold than the default, a shaped reward, float64 instead of float32,
    # 0.0 instead of -0.0 is the same VALUE but the reference's expression yields -0.0: bits are compared
    with pytest.raises(NotImplementedError, match="not the goal-distance reward"):
        her_sampler("future", 4, ref_sparse(0.1), rng=dev)
    with pytest.raises(NotImplementedError, match="not the goal-distance reward"):
        her_sampler("future", 4, lambda ag, g, info: -np.linalg.norm(ag - g, axis=-1) ** 2, reward_type="dense", rng=dev)
    with pytest.raises(NotImplementedError, match="not the goal-distance reward"):
        her_sampler("future", 4, lambda ag, g, info: -(np.linalg.norm(ag - g, axis=-1) > 0.05).astype(np.float64), rng=dev)
    with pytest.raises(NotImplementedError, match="could not be evaluated"):
        her_sampler("future", 4, lambda ag: 0.0, rng=dev)
    # a bound env method: attributes honoured AND checked against what the method really computes
    env = PointMassGoalEnv(seed=1, distance_threshold=0.08)
    assert her_sampler("future", 4, env.compute_reward, rng=dev).distance_threshold == 0.08
    env.distance_threshold = 0.03            # the attribute now lies about ... nothing: the method reads it, still consistent
    assert her_sampler("future", 4, env.compute_reward, rng=dev).distance_threshold == 0.03

    class Lying:
        distance_threshold, reward_type = 0.05, "sparse"

        def compute_reward(self, ag, g, info):
            return -(np.linalg.norm(ag - g, axis=-1) > 0.2).astype(np.float32)

    with pytest.raises(NotImplementedError, match="not the goal-distance reward"):
        her_sampler("future", 4, Lying().compute_reward, rng=dev)


def test_her_sampler_on_host_episode_dict():
    """her_sampler.sample_her_transitions(episode_batch, B) as ddpg_agent._update_normalizer calls it."""
    eps = make_episodes(2, seed=5, mode="walk")
    batch = {"obs": eps[0], "ag": eps[1], "g": eps[2], "actions": eps[3],
             "obs_next": eps[0][:, 1:], "ag_next": eps[1][:, 1:]}
    from oracle.her_replay import sample_her_transitions
    rs = np.random.RandomState(77)
    ref, _ = sample_her_transitions(batch, 100, 0.8, rs)
    dev = fresh_rng(77)
    tr = her_sampler("future", 4, rng=dev).sample_her_transitions(batch, 100)
    for key in KEYS:
        assert np.array_equal(bits(tr[key]), bits(ref[key])), key
    assert state_equal(dev, *rs.get_state()[1:3])


def test_non_future_strategy_never_relabels():
    eps = make_episodes(4, seed=2, mode="walk")
    dev = fresh_rng(1)
    s = her_sampler("final", 4, rng=dev)
    assert s.future_p == 0
    buf = replay_buffer(ENV_PARAMS, 400, s.sample_her_transitions, rng=dev)
    buf.store_episode(eps)
    rs = np.random.RandomState(1)
    st = EpisodeStore(100, 27, 3, 4, 400); st.store_episode(eps, rs)
    ref, _ = st.sample(64, 0, rs)
    tr = buf.sample(64)
    for key in KEYS:
        assert np.array_equal(bits(tr[key]), bits(ref[key])), key


def test_dense_reward_matches_compute_reward_bitwise():
    """reward_type='dense' (compute_reward :89-90, -d in float64): the sampler returns the env's float64 values, the
    learner's float32 reward vector is their narrowing (checked through the update test below)."""
    from oracle.her_replay import compute_reward
    from rl_arm_under_sparse_reward_amd.her import her_sampler
    from rl_arm_under_sparse_reward_amd.replay_buffer import replay_buffer
    rng = fresh_rng(31)
    hs = her_sampler("future", 4, None, distance_threshold=0.05, reward_type="dense", rng=rng)
    assert hs.sq_threshold < 0
    buf = replay_buffer(dict(ENV_PARAMS), 40 * 100, hs.sample_her_transitions, rng=rng)
    buf.store_episode(make_episodes(40, seed=5, mode="walk"))
    tr = buf.sample(512)
    want = compute_reward(tr["ag_next"], tr["g"], reward_type="dense")[:, None]
    assert tr["r"].dtype == np.float64 and tr["r"].shape == (512, 1)
    assert np.array_equal(tr["r"].view(np.uint64), want.view(np.uint64))
    assert (tr["r"] <= 0).all() and (tr["r"] < 0).any()


def _random_shapes():
    rs = np.random.RandomState(2024)
    cases = [(1, 1, 1, 3, 1, 1, 4), (1, 2, 7, 5, 2, 3, 0), (3, 1, 64, 9, 3, 4, 8)]      # degenerate N / T / B, no relabelling
    for _ in range(9):
        cases.append((int(rs.randint(1, 40)), int(rs.randint(1, 60)), int(rs.randint(1, 700)), int(rs.randint(1, 65)),
                      int(rs.randint(1, 8)), int(rs.randint(1, 9)), int(rs.choice([0, 1, 4, 8, 100]))))
    return cases


@pytest.mark.parametrize("n,T,B,od,gd,ad,k", _random_shapes())
def test_sampler_matches_oracle_on_random_shapes(n, T, B, od, gd, ad, k):
    """Seeded sweep over buffer / episode / batch / row shapes, incl. one episode, one timestep, one transition, no
    relabelling (k = 0) and almost-always relabelling (k = 100): whole sample dict, indices and RNG position bit for bit."""
    rs0 = np.random.RandomState(n * 1000 + T)
    eps = [rs0.normal(size=(n, T + 1, od)), rs0.normal(0, 0.05, size=(n, T + 1, gd)), rs0.normal(0, 0.05, size=(n, T, gd)),
           rs0.uniform(-1, 1, size=(n, T, ad))]
    fp = future_probability("future", k) if k else 0.0
    st = EpisodeStore(T, od, gd, ad, n * T)
    rs = np.random.RandomState(77)
    st.store_episode(eps, rs)
    dev = fresh_rng(77)
    buf = DeviceEpisodeBuffer(n, T, od, gd, ad)
    buf.store(dev, eps)
    for _ in range(2):
        ref, ridx = st.sample(B, fp, rs)
        tr, idx = buf.sample(dev, B, fp, squared_threshold(0.05), with_indices=True)
        for key in KEYS:
            assert tr[key].shape == ref[key].shape and np.array_equal(bits(tr[key]), bits(ref[key])), key
        for key in ("e", "t", "future_t", "her"):
            assert np.array_equal(idx[key], ridx[key]), key
    assert state_equal(dev, *rs.get_state()[1:3])


@pytest.mark.parametrize("case", range(8))
def test_storage_policy_matches_oracle_on_random_store_sequences(case):
    """_get_storage_idx (replay_buffer.py:57-71) through all three branches -- append, tail + random overflow, all random
    (incl. batches larger than the buffer and repeated slots) -- on random buffer sizes and store sequences: buffer
    contents, counters and the position of the shared random stream after every store."""
    rs0 = np.random.RandomState(900 + case)
    size = int(rs0.randint(1, 24))
    T, od, gd, ad = int(rs0.randint(1, 6)), int(rs0.randint(1, 5)), int(rs0.randint(1, 4)), int(rs0.randint(1, 4))
    dev, rs = fresh_rng(31 + case), np.random.RandomState(31 + case)
    buf = DeviceEpisodeBuffer(size, T, od, gd, ad)
    st = EpisodeStore(T, od, gd, ad, size * T)
    first = True
    for _ in range(int(rs0.randint(3, 12))):
        inc = int(rs0.randint(1, 2 * size + 2))
        if first and inc > size:
            inc = size                  # an empty buffer cannot take more than it holds (randint(0, 0) raises in both)
        first = False
        eps = [rs0.normal(size=(inc, T + 1, od)), rs0.normal(size=(inc, T + 1, gd)), rs0.normal(size=(inc, T, gd)),
               rs0.normal(size=(inc, T, ad))]
        buf.store(dev, eps)
        st.store_episode(eps, rs)
        assert buf.info()[1] == st.current_size and buf.info()[2] == st.n_transitions_stored
        assert state_equal(dev, *rs.get_state()[1:3])
        cur = st.current_size
        for key in ("obs", "ag", "g", "actions"):
            assert np.array_equal(buf.read(key, 0, cur), st.buffers[key][:cur]), key


def test_wide_goals_are_refused_not_relabelled_with_other_bits():
    """the device sums the squared goal distance in index order, numpy's reduction goes pairwise from 8 contiguous
    elements on -- bit-exact rewards are claimed for goal_dim < 8 (the reference has 3) and anything wider is refused up front,
    with or without a reward callable."""
    from rl_arm_under_sparse_reward_amd.her import her_sampler
    for gd in (8, 16):
        with pytest.raises(NotImplementedError, match="fewer than 8 goal components"):
            her_sampler("future", 4, None, goal_dim=gd)
    her_sampler("future", 4, None, goal_dim=7)


# ---- device-output fused sampler (hp_buffer_sample_dev; SURVEY 8b `hp_sample(... out device ptrs x, x', a, r)`) -------------
def _primed_normalizers(eps, seed=9):
    """Oracle normalizers with non-trivial statistics + device normalizers holding exactly the same state."""
    from oracle.running_norm import RunningNorm
    from rl_arm_under_sparse_reward_amd.normalizer import normalizer
    from gpu_common import ctx

    rs = np.random.RandomState(seed)
    on, gn = RunningNorm(27, default_clip_range=5), RunningNorm(3, default_clip_range=5)
    o_dev, g_dev = normalizer(27, default_clip_range=5, ctx=ctx()), normalizer(3, default_clip_range=5, ctx=ctx())
    rows = rs.randint(0, eps[0].shape[0], 100)
    o = eps[0][rows, rs.randint(0, 100, 100)] * 3.0 + 0.25        # wide enough that the +-5 clip bites on some columns
    g = eps[2][rows, rs.randint(0, 100, 100)]
    for a, b in ((on, o_dev), (gn, g_dev)):
        v = o if a is on else g
        a.update(v); b.update(v)
        a.recompute_stats(); b.recompute_stats()
    assert np.array_equal(bits(on.mean), bits(o_dev.mean)) and np.array_equal(bits(gn.std), bits(g_dev.std))
    return on, gn, o_dev, g_dev


def _assert_minibatch_equal(got, tr, on, gn, clip_obs=200):
    from oracle.ddpg_update import minibatch_tensors

    x, xn, a, r = (t.numpy() for t in minibatch_tensors(tr, on, gn, clip_obs))
    for key, ref in (("x", x), ("x_next", xn), ("actions", a), ("r", r)):
        dev = got[key].cpu().numpy()
        assert dev.dtype == np.float32 and dev.shape == ref.shape, (key, dev.shape, ref.shape)
        assert np.array_equal(bits(dev), bits(ref)), key


def test_sample_device_golden_bitwise():
    """The reference-generated HER goldens through ddpg_agent.py:227-243's preprocessing (oracle.minibatch_tensors) ==
    hp_buffer_sample_dev's device tensors, bit for bit, and the random stream ends where the reference's did."""
    g = load_golden("her_sample.npz")
    for tag in g["cases"]:
        tag = str(tag)
        n, B, k, seed, dseed = (int(x) for x in g[tag + "_meta"])
        eps = make_episodes(n, seed=dseed, mode=str(g[tag + "_mode"]))
        on, gn, o_dev, g_dev = _primed_normalizers(eps)
        dev = fresh_rng(seed)
        sampler = her_sampler("future", k, rng=dev)
        buf = replay_buffer(ENV_PARAMS, n * 100, sampler.sample_her_transitions, rng=dev)
        buf.store_episode(eps)
        got = buf.sample_device(B, o_dev, g_dev, clip_obs=200)
        assert set(got) == {"x", "x_next", "actions", "r"} and all(t.is_cuda for t in got.values())
        _assert_minibatch_equal(got, {key: g[f"{tag}_{key}"] for key in KEYS}, on, gn)
        assert state_equal(dev, g[tag + "_key"], g[tag + "_pos"]), tag


@pytest.mark.parametrize("n,B,k,mode,clip_obs", [(5000, 256, 4, "iid", 200), (5000, 4096, 8, "walk", 200),
                                                  (37, 1001, 4, "walk", 0.4), (5000, 65536, 4, "iid", 200)])
def test_sample_device_matches_oracle_at_baseline_sizes(n, B, k, mode, clip_obs):
    """5000-episode shard at batch 256 / 4096 (BASELINE configs 2, 5), a ragged batch with a biting _preproc_og clip, and a
    batch large enough that the kernel grid-strides; indices, rewards and float32 rows bit-equal, three draws in a row
    interleaved with the host-output sampler on the same stream."""
    eps = make_episodes(n, seed=1, mode=mode)
    fp = future_probability("future", k)
    st = EpisodeStore(100, 27, 3, 4, n * 100)
    rs = np.random.RandomState(125)
    st.store_episode(eps, rs)
    on, gn, o_dev, g_dev = _primed_normalizers(eps)
    dev = fresh_rng(125)
    buf = DeviceEpisodeBuffer(n, 100, 27, 3, 4)
    buf.store(dev, eps)
    for i in range(3):
        ref, ridx = st.sample(B, fp, rs)
        if i == 1:       # the host-output sampler draws from the same stream in between
            tr = buf.sample(dev, B, fp, squared_threshold(0.05))
            assert np.array_equal(bits(tr["obs"]), bits(ref["obs"]))
            continue
        got, idx = buf.sample_device(dev, o_dev, g_dev, B, fp, squared_threshold(0.05), clip_obs, with_indices=True)
        _assert_minibatch_equal(got, ref, on, gn, clip_obs)
        for key in ("e", "t", "future_t"):
            assert np.array_equal(idx[key].cpu().numpy(), ridx[key]), key
        assert np.array_equal(idx["her"].cpu().numpy().astype(bool), ridx["her"])
    assert state_equal(dev, *rs.get_state()[1:3])
    assert set(np.unique(got["r"].cpu().numpy().view(np.uint32))) <= {0x80000000, 0xBF800000}


@pytest.mark.parametrize("od,gd,ad,B", [(12, 5, 3, 700), (7, 2, 2, 20000), (40, 3, 4, 513), (27, 3, 4, 19999), (30, 1, 4, 300)])
def test_sample_device_other_shapes_take_the_right_kernel_and_stay_bit_exact(od, gd, ad, B):
    """Round 6: the fused sampler packs a transition into 32 lanes of 16-byte loads (k_gather_fused2) when obs + ceil(goal / 2) +
    ceil(act / 2) <= 32 -- the reference's (27, 3, 4) -- and keeps the one-element-per-lane kernel for anything else.  Odd goal /
    action widths (a unit's tail loads [len - 2, len - 1], never past the row), a shape too wide for 32 lanes, a 1-wide goal, and
    the reference's shape at a ragged grid-striding batch: all bit-equal to the oracle, rewards and indices included."""
    from oracle.running_norm import RunningNorm
    from rl_arm_under_sparse_reward_amd.normalizer import normalizer
    from gpu_common import ctx

    n, T = 23, 100
    rs0 = np.random.RandomState(77)
    obs = rs0.uniform(-1, 1, (n, T + 1, od))
    ag = rs0.uniform(0, 0.3, (n, T + 1, gd))
    ag[:, 1::7] = ag[:, 0:-1:7][:, :ag[:, 1::7].shape[1]]          # some exact successes beside the relabelled ones
    g = np.repeat(rs0.uniform(0, 0.3, (n, 1, gd)), T, axis=1)
    act = rs0.uniform(-0.5, 0.5, (n, T, ad))
    eps = [obs, ag, g, act]
    on, gn = RunningNorm(od, default_clip_range=5), RunningNorm(gd, default_clip_range=5)
    o_dev, g_dev = normalizer(od, default_clip_range=5, ctx=ctx()), normalizer(gd, default_clip_range=5, ctx=ctx())
    for a_, b_, v in ((on, o_dev, obs[:, 3] * 2.0 + 0.1), (gn, g_dev, ag[:, 5])):
        a_.update(v); b_.update(v)
        a_.recompute_stats(); b_.recompute_stats()
    st = EpisodeStore(T, od, gd, ad, n * T)
    rs = np.random.RandomState(31)
    st.store_episode(eps, rs)
    dev = fresh_rng(31)
    buf = DeviceEpisodeBuffer(n, T, od, gd, ad)
    buf.store(dev, eps)
    for _ in range(2):
        ref, ridx = st.sample(B, 0.8, rs)
        got, idx = buf.sample_device(dev, o_dev, g_dev, B, 0.8, squared_threshold(0.05), 0.7, with_indices=True)
        from oracle.ddpg_update import minibatch_tensors
        x, xn, a, r = (t.numpy() for t in minibatch_tensors(ref, on, gn, 0.7))
        for key, want in (("x", x), ("x_next", xn), ("actions", a), ("r", r)):
            assert np.array_equal(bits(got[key].cpu().numpy()), bits(want)), key
        for key in ("e", "t", "future_t"):
            assert np.array_equal(idx[key].cpu().numpy(), ridx[key]), key
    assert state_equal(dev, *rs.get_state()[1:3])
    assert len(set(np.unique(got["r"].cpu().numpy().view(np.uint32)))) == 2        # both reward values occur


def _f32_rounded(tr):
    out = dict(tr)
    for key in ("obs", "obs_next", "actions"):
        out[key] = tr[key].astype(np.float32).astype(np.float64)
    return out


@pytest.mark.parametrize("n,B,k,mode,clip_obs", [(37, 1001, 4, "walk", 0.4), (5000, 256, 4, "iid", 200), (5000, 65536, 8, "walk", 200)])
def test_sample_device_f32_rows_throughput_mode(n, B, k, mode, clip_obs):
    """SURVEY 8b `storage_dtype = fp32` as an opt-in mirror (hp_buffer_enable_f32_rows / hp_buffer_sample_dev_f32): same stream,
    same draws; indices, relabelled goals, rewards, the goal columns of x / x_next and the actions BIT-identical to the float64
    path (the goals stay float64 in the mirror); the observation columns are exactly the reference arithmetic applied to
    float32-rounded observations.  The mirror follows later stores (random-slot overwrites of a full buffer included)."""
    eps = make_episodes(n, seed=1, mode=mode)
    fp = future_probability("future", k)
    st = EpisodeStore(100, 27, 3, 4, n * 100)
    rs = np.random.RandomState(125)
    st.store_episode(eps, rs)
    on, gn, o_dev, g_dev = _primed_normalizers(eps)
    dev = fresh_rng(125)
    buf = DeviceEpisodeBuffer(n, 100, 27, 3, 4)
    buf.store(dev, eps)
    with pytest.raises(Exception, match="hp_buffer_enable_f32_rows first"):
        buf.sample_device(dev, o_dev, g_dev, 8, fp, squared_threshold(0.05), clip_obs, f32_rows=True)
    buf.enable_f32_rows()                               # built from what the buffer holds
    for i in range(3):
        if i == 2:                                      # a full buffer: random slots overwritten, the mirror must follow
            more = make_episodes(3, seed=77, mode="walk")
            st.store_episode(more, rs)
            buf.store(dev, more)
        ref, ridx = st.sample(B, fp, rs)
        got, idx = buf.sample_device(dev, o_dev, g_dev, B, fp, squared_threshold(0.05), clip_obs, with_indices=True, f32_rows=True)
        _assert_minibatch_equal(got, _f32_rounded(ref), on, gn, clip_obs)
        for key in ("e", "t", "future_t"):
            assert np.array_equal(idx[key].cpu().numpy(), ridx[key]), key
        # against the float64 path itself: goal columns, actions and rewards are the SAME bits
        from oracle.ddpg_update import minibatch_tensors
        x64, xn64, a64, r64 = (t.numpy() for t in minibatch_tensors(ref, on, gn, clip_obs))
        assert np.array_equal(bits(got["x"].cpu().numpy()[:, 27:]), bits(x64[:, 27:]))
        assert np.array_equal(bits(got["x_next"].cpu().numpy()[:, 27:]), bits(xn64[:, 27:]))
        assert np.array_equal(bits(got["actions"].cpu().numpy()), bits(a64)) and np.array_equal(bits(got["r"].cpu().numpy()), bits(r64))
        assert np.max(np.abs(got["x"].cpu().numpy() - x64)) <= 2e-6      # float32 rounding of an observation, through a std >= 0.01... small
    assert state_equal(dev, *rs.get_state()[1:3])


def test_f32_rows_follow_the_cycle_graph_and_keep_losses_within_the_bar():
    """The mirror behind hp_agent_train_cycle's in-launch scatter (k_cycle_open -> k_pack_rows inside the cycle graph), and the
    north-star bar on what the mode changes: an update on a throughput-mode minibatch has both losses within 1e-5 (relative) of the
    same update on the float64-row minibatch."""
    import torch
    from oracle import ddpg_update as oupd
    from rl_arm_under_sparse_reward_amd.arguments import Args
    from rl_arm_under_sparse_reward_amd.ddpg_agent import ddpg_agent
    from gpu_common import ctx

    torch.manual_seed(0)
    rng = fresh_rng(11)
    agent = ddpg_agent(Args(batch_size=256, buffer_size=40 * 100), None, dict(ENV_PARAMS), ctx=ctx(), rng=rng)
    agent.buffer.store_episode(make_episodes(40, seed=5, mode="walk"))      # full: the cycles below overwrite random slots
    agent.buffer.enable_f32_rows()
    for c in range(3):
        agent.train_cycle(make_episodes(2, seed=60 + c, mode="walk"), 4)
    state = rng.get_state()
    a = agent.buffer.sample_device(512, agent.o_norm, agent.g_norm, clip_obs=200)
    rng.set_state(state)                                 # the same draws again, through the mirror
    b = agent.buffer.sample_device(512, agent.o_norm, agent.g_norm, clip_obs=200, f32_rows=True)
    for key in ("actions", "r"):
        assert np.array_equal(bits(a[key].cpu().numpy()), bits(b[key].cpu().numpy())), key
    for key in ("x", "x_next"):
        xa, xb = a[key].cpu().numpy(), b[key].cpu().numpy()
        assert np.array_equal(bits(xa[:, 27:]), bits(xb[:, 27:]))
        assert 0 < np.max(np.abs(xa - xb)) <= 2e-6, key          # rounded observations: different bits, tiny differences
    # the episodes stored by the cycles really are in the mirror: the float64 rows, rounded, reproduce it bitwise
    obs64 = agent.buffer.buffers["obs"]
    row = np.zeros((40, 101, 32), np.float32)
    row[:, :, :27] = obs64.astype(np.float32)
    row[:, :100, 27:31] = agent.buffer.buffers["actions"].astype(np.float32)
    on, gn = agent.o_norm, agent.g_norm
    # one update from identical networks on both minibatches (CPU oracle learner)
    a0, c0 = oupd.init_actor(27, 3, 4, 0), oupd.init_critic(27, 3, 4, 1)
    la = oupd.DDPGLearner({k: v.clone() for k, v in a0.items()}, {k: v.clone() for k, v in c0.items()}).update(
        *(a[k].cpu() for k in ("x", "x_next", "actions", "r")))
    lb = oupd.DDPGLearner({k: v.clone() for k, v in a0.items()}, {k: v.clone() for k, v in c0.items()}).update(
        *(b[k].cpu() for k in ("x", "x_next", "actions", "r")))
    for name in ("actor_loss", "critic_loss"):
        assert abs(la[name] - lb[name]) <= 1e-5 * max(abs(la[name]), 1e-6), (name, la[name], lb[name])


@pytest.mark.parametrize("n,B,k,f32_rows", [(37, 1001, 4, False), (5000, 256, 4, False), (5000, 70000, 8, False), (5000, 20001, 4, True)])
def test_sample_device_fast_draw_matches_its_oracle_twin(n, B, k, f32_rows):
    """SURVEY 8b `rng_mode` = Philox as an opt-in (hp_buffer_sample_dev_fast): the index draw inside the gather kernel, Philox4x32-10
    keyed by (seed, call, transition) -- pinned on the CPU by Random123's known-answer vectors (tests/test_oracle_rng.py) -- must
    give exactly the indices of the numpy twin, and from them the same float32 minibatch as every other path; the MT19937 stream
    is left untouched; the same (seed, call) gives the same minibatch, the next call another."""
    eps = make_episodes(n, seed=1, mode="walk")
    fp = future_probability("future", k)
    st = EpisodeStore(100, 27, 3, 4, n * 100)
    rs = np.random.RandomState(125)
    st.store_episode(eps, rs)
    on, gn, o_dev, g_dev = _primed_normalizers(eps)
    dev = fresh_rng(125)
    buf = DeviceEpisodeBuffer(n, 100, 27, 3, 4)
    buf.store(dev, eps)
    if f32_rows:
        buf.enable_f32_rows()
    state = dev.get_state()
    seed = 125 + (7 << 32)                       # both key words in use
    first = None
    for call in (0, 1, (1 << 33) + 5):
        ref, ridx = st.sample_fast(B, fp, seed, call)
        got, idx = buf.sample_device(dev, o_dev, g_dev, B, fp, squared_threshold(0.05), 200, with_indices=True, f32_rows=f32_rows,
                                     fast=(seed, call))
        for key in ("e", "t", "future_t"):
            assert np.array_equal(idx[key].cpu().numpy(), ridx[key]), (key, call)
        assert np.array_equal(idx["her"].cpu().numpy().astype(bool), ridx["her"])
        _assert_minibatch_equal(got, _f32_rounded(ref) if f32_rows else ref, on, gn, 200)
        if first is None:
            first = got["x"].cpu().numpy().copy()
        else:
            assert not np.array_equal(first, got["x"].cpu().numpy())
    again = buf.sample_device(dev, o_dev, g_dev, B, fp, squared_threshold(0.05), 200, f32_rows=f32_rows, fast=(seed, 0))
    assert np.array_equal(bits(again["x"].cpu().numpy()), bits(first))
    assert state_equal(dev, state[1], state[2])           # the reference's stream was not consumed


def _device_fuzz_shapes():
    rs = np.random.RandomState(606)
    cases = [(1, 1, 1, 27, 3, 4), (1, 2, 33, 1, 1, 1), (2, 1, 4097, 28, 4, 4), (5, 3, 16385, 31, 1, 2)]   # one episode / one timestep / the
    for _ in range(10):                                                                              # widest shapes of each kernel
        cases.append((int(rs.randint(1, 30)), int(rs.randint(1, 50)), int(rs.choice([1, 7, 64, 255, 1000, 4099, 20000])),
                      int(rs.randint(1, 45)), int(rs.randint(1, 7)), int(rs.randint(1, 8))))
    return cases


@pytest.mark.parametrize("n,T,B,od,gd,ad", _device_fuzz_shapes())
def test_sample_device_fuzz_over_shapes_and_modes(n, T, B, od, gd, ad):
    """Seeded sweep of hp_buffer_sample_dev / _f32 / _fast over buffer, episode, batch and row shapes (one episode, one timestep, one
    transition, batches on both sides of the kernels' in-flight switch, rows too wide for the packed kernels): every mode the shape
    admits against the oracle -- MT19937 draw on float64 rows, on the float32 mirror where a row fits one line, Philox draw."""
    from oracle.running_norm import RunningNorm
    from rl_arm_under_sparse_reward_amd.normalizer import normalizer
    from gpu_common import ctx

    rs0 = np.random.RandomState(1000 * n + 10 * T + od)
    obs = rs0.uniform(-1, 1, (n, T + 1, od))
    ag = rs0.uniform(0, 0.2, (n, T + 1, gd))
    g = np.repeat(rs0.uniform(0, 0.2, (n, 1, gd)), T, axis=1)
    act = rs0.uniform(-0.5, 0.5, (n, T, ad))
    eps = [obs, ag, g, act]
    on, gn = RunningNorm(od, default_clip_range=5), RunningNorm(gd, default_clip_range=5)
    o_dev, g_dev = normalizer(od, default_clip_range=5, ctx=ctx()), normalizer(gd, default_clip_range=5, ctx=ctx())
    for a_, b_, v in ((on, o_dev, obs[:, 0] * 2.5 - 0.2), (gn, g_dev, ag[:, -1])):
        a_.update(v); b_.update(v)
        a_.recompute_stats(); b_.recompute_stats()
    st = EpisodeStore(T, od, gd, ad, n * T)
    rs = np.random.RandomState(9)
    st.store_episode(eps, rs)
    dev = fresh_rng(9)
    buf = DeviceEpisodeBuffer(n, T, od, gd, ad)
    buf.store(dev, eps)
    sq, clip = squared_threshold(0.05), 0.9
    ref, ridx = st.sample(B, 0.8, rs)
    got, idx = buf.sample_device(dev, o_dev, g_dev, B, 0.8, sq, clip, with_indices=True)
    _assert_minibatch_equal(got, ref, on, gn, clip)
    for key in ("e", "t", "future_t"):
        assert np.array_equal(idx[key].cpu().numpy(), ridx[key]), key
    mirror = od + ad <= 32 and od <= 28 and gd <= 4
    if mirror:
        buf.enable_f32_rows()
        ref, ridx = st.sample(B, 0.8, rs)
        got, idx = buf.sample_device(dev, o_dev, g_dev, B, 0.8, sq, clip, with_indices=True, f32_rows=True)
        _assert_minibatch_equal(got, _f32_rounded(ref), on, gn, clip)
        assert np.array_equal(idx["future_t"].cpu().numpy(), ridx["future_t"])
    else:
        with pytest.raises(Exception):
            buf.enable_f32_rows()
    assert state_equal(dev, *rs.get_state()[1:3])
    packed = od + (gd + 1) // 2 + (ad + 1) // 2 <= 32 and gd >= 2 and ad >= 2     # the in-kernel draw lives in the 32-lane kernels only
    if not packed:
        with pytest.raises(ValueError, match="hp_buffer_sample_dev_fast: needs obs_dim"):
            buf.sample_device(dev, o_dev, g_dev, B, 0.8, sq, clip, fast=(31337, 4))
    for f32_rows in ((False, True) if mirror else (False,)) if packed else ():
        ref, ridx = st.sample_fast(B, 0.8, 31337, 4)
        got, idx = buf.sample_device(dev, o_dev, g_dev, B, 0.8, sq, clip, with_indices=True, f32_rows=f32_rows, fast=(31337, 4))
        for key in ("e", "t", "future_t"):
            assert np.array_equal(idx[key].cpu().numpy(), ridx[key]), (key, f32_rows)
        _assert_minibatch_equal(got, _f32_rounded(ref) if f32_rows else ref, on, gn, clip)
    assert state_equal(dev, *rs.get_state()[1:3])


def test_sample_device_dense_reward_partial_outputs_and_errors():
    """Dense reward (compute_reward :89-90 narrowed to float32 as ddpg_agent.py:243 does), NULL outputs, and the reference's
    error on an empty buffer."""
    import ctypes as C
    import torch
    from rl_arm_under_sparse_reward_amd import _lib

    eps = make_episodes(12, seed=4, mode="walk")
    on, gn, o_dev, g_dev = _primed_normalizers(eps)
    st = EpisodeStore(100, 27, 3, 4, 1200)
    rs = np.random.RandomState(7)
    st.store_episode(eps, rs)
    dev = fresh_rng(7)
    buf = DeviceEpisodeBuffer(12, 100, 27, 3, 4)
    with pytest.raises(ValueError, match="high <= 0"):
        buf.sample_device(dev, o_dev, g_dev, 8, 0.8, squared_threshold(0.05), 200)
    buf.store(dev, eps)
    ref, _ = st.sample(300, 0.8, rs, reward_fn=lambda a, b: compute_reward(a, b, reward_type="dense"))
    got = buf.sample_device(dev, o_dev, g_dev, 300, 0.8, -1.0, 200)
    assert ref["r"].dtype == np.float64            # compute_reward :89-90; torch.tensor(.., float32) narrows it (:243)
    _assert_minibatch_equal(got, ref, on, gn)
    # only r requested
    r = torch.empty(64, dtype=torch.float32, device="cuda")
    o = _lib.SampleDevOut()
    o.r = r.data_ptr()
    lib = buf.lib
    _lib.check(lib.hp_buffer_sample_dev(buf.h, dev.h, o_dev.h, g_dev.h, 64, 0.8, squared_threshold(0.05), 200.0, C.byref(o)))
    torch.cuda.synchronize()
    assert set(np.unique(r.cpu().numpy().view(np.uint32))) <= {0x80000000, 0xBF800000}
    with pytest.raises(ValueError, match="normalizer sizes"):
        buf.sample_device(dev, g_dev, o_dev, 8, 0.8, squared_threshold(0.05), 200)
