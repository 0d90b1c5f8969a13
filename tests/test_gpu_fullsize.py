"""BASELINE.json configurations 2 and 3 end to end at their real sizes, through the path bench.py times:
hp_agent_train_cycle = store (random-slot overwrite of a FULL 5000-episode buffer) -> normalizer refresh -> 40 x
(HER sample fused into k_fb_slab8 via s8_gather / s8_gather_ahead over the 149 MB shard, update, Adam) -> polyak.
The oracle is fed the same stream; sampled indices are checked through the RNG state (bit-exact), losses within the
north-star 1e-5 relative for the first update and the envelope 1e-5 x 1.3^i (capped at 3e-3) along the chained trajectory (two correct fp32
implementations separate slowly; same bound as tests/test_gpu_update.py).
  config 2: push, buffer 5e5, batch 256, replay_k 4
  config 3: add_demo (first 1000 episodes from a 1000-episode demo .npz in the get_demo_data schema), batch 1024
  config 5 per GPU: batch 512 (global 4096 / 8), replay_k 8, 8 fresh episodes per cycle (64 envs / 8 GPUs)"""
import numpy as np
import pytest
import torch

from conftest import bits
from gpu_common import ENV_PARAMS, fresh_rng, state_equal
from oracle import ddpg_update as oupd
from oracle.her_replay import EpisodeStore, future_probability
from oracle.running_norm import RunningNorm, update_normalizers
from rl_arm_under_sparse_reward_amd.arguments import Args
from rl_arm_under_sparse_reward_amd.ddpg_agent import NET_ACTOR, NET_CRITIC, ddpg_agent
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes, write_demo_npz

pytestmark = pytest.mark.gpu
N_EPISODES = 5000          # buffer_size 5e5 / T 100 (replay_buffer.py:16)


@pytest.mark.parametrize("batch,k,n_demo,n_fresh", [(256, 4, 0, 2), (1024, 4, 1000, 2), (512, 8, 0, 8)],
                         ids=["config2_b256", "config3_demo_b1024", "config5_per_gpu_b512_k8"])
def test_train_cycles_on_the_full_buffer_track_the_oracle(batch, k, n_demo, n_fresh, tmp_path):
    torch.set_num_threads(8)
    seed, n_cycles, n_batches = 125, 2, 40
    kw = {}
    demo_eps = None
    if n_demo:
        path = str(tmp_path / f"bmirobot_{n_demo}_pick_demo.npz")
        demo_eps = list(write_demo_npz(path, n_episodes=n_demo, seed=7))
        kw = dict(add_demo=True, demo_name=path)
    args = Args(batch_size=batch, buffer_size=N_EPISODES * 100, replay_k=k, **kw)
    rng = fresh_rng(seed)
    torch.manual_seed(0)
    agent = ddpg_agent(args, None, dict(ENV_PARAMS), rng=rng)
    assert agent.buffer.size == N_EPISODES and agent.buffer.current_size == n_demo
    a0 = {kk: v.detach().clone() for kk, v in agent.actor_network.state_dict().items()}
    c0 = {kk: v.detach().clone() for kk, v in agent.critic_network.state_dict().items()}
    rest = make_episodes(N_EPISODES - n_demo, seed=1)
    agent.buffer.store_episode(rest)
    assert agent.buffer.current_size == N_EPISODES
    # oracle twin
    rs = np.random.RandomState(seed)
    st = EpisodeStore(100, 27, 3, 4, N_EPISODES * 100)
    if n_demo:
        st.store_episode(demo_eps, rs)
    st.store_episode(rest, rs)
    fp = future_probability("future", k)
    on, gn = RunningNorm(27, default_clip_range=5), RunningNorm(3, default_clip_range=5)
    learner = oupd.DDPGLearner(a0, c0)
    for cycle in range(n_cycles):
        eps = make_episodes(n_fresh, seed=100 + cycle, mode="walk")
        agent.train_cycle(eps, n_batches)
        slots = st.store_episode(eps, rs)                      # overflow branch: randint(0, size, n_fresh)
        assert np.array_equal(agent.buffer._dev.last_slots(n_fresh), slots)
        update_normalizers(on, gn, eps, fp, rs)
        got = agent.last_losses(n_batches)
        for i in range(n_batches):
            tr, _ = st.sample(batch, fp, rs)
            res = learner.update(*oupd.minibatch_tensors(tr, on, gn))
            tol = min(1e-5 * 1.3 ** (cycle * n_batches + i), 3e-3)   # chained comparison: see test_gpu_update.py
            for j, name in enumerate(("actor_loss", "critic_loss")):
                assert abs(got[i, j] - res[name]) <= tol * max(abs(res[name]), 1e-2), (cycle, i, name, got[i], res[name])
        learner.soft_update()
        assert state_equal(rng, *rs.get_state()[1:3]), cycle    # every index draw consumed the reference's words
        assert np.array_equal(bits(agent.o_norm.mean), bits(on.mean)) and np.array_equal(bits(agent.g_norm.std), bits(gn.std))
    # the overwritten slots hold the fresh episodes; a demo episode that was not overwritten is intact
    last = make_episodes(n_fresh, seed=100 + n_cycles - 1, mode="walk")
    stored = agent.buffer._dev.read("obs", int(slots[-1]), 1)
    assert np.array_equal(stored[0], last[0][-1])
    if n_demo:
        keep = next(i for i in range(n_demo) if i not in set(int(s) for s in slots))
        assert np.array_equal(agent.buffer._dev.read("ag", keep, 1)[0], demo_eps[1][keep])
    rel = np.linalg.norm(agent._get_flat(NET_CRITIC).astype(np.float64) - learner.flat("critic")) / np.linalg.norm(
        learner.flat("critic") - oupd.flatten(list(c0.values())))
    assert rel <= 0.2, rel
    assert np.all(np.isfinite(agent._get_flat(NET_ACTOR)))


def test_dict_sampler_and_fused_gather_agree_at_full_size():
    """The same draws through both samplers on the 5000-episode buffer: replay_buffer.sample (k_gather_dict) returns
    the oracle's transitions bit for bit, and the fused in-kernel gather of an update from the same RNG state yields
    losses equal to an update on the explicitly normalised dict batch."""
    seed, B, k = 9, 256, 4
    args = Args(batch_size=B, buffer_size=N_EPISODES * 100, replay_k=k)
    rng = fresh_rng(seed)
    torch.manual_seed(1)
    agent = ddpg_agent(args, None, dict(ENV_PARAMS), rng=rng)
    eps = make_episodes(N_EPISODES, seed=1)
    agent.buffer.store_episode(eps)
    agent._update_normalizer([a[:2] for a in eps])
    st8 = rng.get_state()
    tr = agent.buffer.sample(B)
    rs = np.random.RandomState(0)
    rs.set_state(st8)
    st = EpisodeStore(100, 27, 3, 4, N_EPISODES * 100)
    st.store_episode(eps, np.random.RandomState(0))
    want, _ = st.sample(B, future_probability("future", k), rs)
    for kk in want:
        assert np.array_equal(bits(tr[kk]), bits(want[kk])), kk
    # fused path from the same stream position
    p_a, p_c = agent._get_flat(NET_ACTOR), agent._get_flat(NET_CRITIC)
    rng.set_state(st8)
    agent._update_network(1)
    fused = agent.last_losses(1)[0]
    agent._set_flat(NET_ACTOR, p_a); agent._set_flat(NET_CRITIC, p_c)
    o, g = agent._preproc_og(tr["obs"], tr["g"])
    on_, _ = agent._preproc_og(tr["obs_next"], tr["g"])
    x = np.concatenate([agent.o_norm.normalize(o), agent.g_norm.normalize(g)], axis=1).astype(np.float32)
    xn = np.concatenate([agent.o_norm.normalize(on_), agent.g_norm.normalize(g)], axis=1).astype(np.float32)
    # targets were not touched by one update: the explicit minibatch update starts from the same state
    la, lc = agent.update_on_minibatch(x, xn, tr["actions"].astype(np.float32), tr["r"].astype(np.float32))
    assert np.float32(la) == fused[0] and np.float32(lc) == fused[1], (la, lc, fused)
