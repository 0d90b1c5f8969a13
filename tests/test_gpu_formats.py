"""GPU half of the format / reward pins (CPU half: tests/test_formats_cpu.py): files written by the reference itself
go through the C ABI and must come out as the reference's own reader / functions produce them."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bits, load_golden
from gpu_common import ENV_PARAMS, ctx, fresh_rng
from oracle.her_replay import EpisodeStore, compute_reward, future_probability
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.arguments import Args
from rl_arm_under_sparse_reward_amd.ddpg_agent import NET_ACTOR, ddpg_agent
from rl_arm_under_sparse_reward_amd.goal_env import GoalDistanceReward
from rl_arm_under_sparse_reward_amd.her import her_sampler
from rl_arm_under_sparse_reward_amd.replay_buffer import replay_buffer
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes

pytestmark = pytest.mark.gpu
REF_CKPT = os.path.join(GOLDEN, "ref_checkpoint_model.pt")


def _agent(**kw):
    args = Args(batch_size=64, buffer_size=1600, **kw)
    return ddpg_agent(args, None, dict(ENV_PARAMS), rng=fresh_rng(5))


def test_load_reference_written_checkpoint_and_act_like_demo_push():
    """ddpg_agent.py:158-161 file -> load_checkpoint -> hp_agent_act == the reference reader's actions (demo_push.py)."""
    pr = load_golden("ref_checkpoint_probe.npz")
    agent = _agent()
    agent.load_checkpoint(REF_CKPT)
    for nm, a in (("o_mean", agent.o_norm.mean), ("o_std", agent.o_norm.std), ("g_mean", agent.g_norm.mean),
                  ("g_std", agent.g_norm.std)):
        assert np.array_equal(bits(a), bits(pr[nm])), nm
    model = torch.load(REF_CKPT, map_location="cpu", weights_only=False)[4]
    flat = np.concatenate([model[k].numpy().ravel() for k in model])
    assert np.array_equal(agent._get_flat(NET_ACTOR), flat)
    got = agent.act(pr["probe_obs"], pr["probe_g"], clip_obs=float(pr["clip_obs"]))
    assert got.dtype == np.float32 and got.shape == pr["actions"].shape
    assert np.allclose(got, pr["actions"], rtol=1e-5, atol=1e-6), float(np.abs(got - pr["actions"]).max())
    # the float32 network inputs behind it are bit-identical to process_inputs' (same float64 normalisation)
    x = agent.o_norm.normalize(np.clip(pr["probe_obs"], -200, 200))
    assert np.array_equal(x.astype(np.float32), pr["inputs"][:, :27])
    # actor.forward on the reference's inputs as well (models.py:19-26)
    got2 = agent.actor_network(torch.from_numpy(pr["inputs"])).numpy()
    assert np.allclose(got2, pr["actions"], rtol=1e-5, atol=1e-6)


def test_our_checkpoint_has_the_reference_files_structure(tmp_path):
    ref = torch.load(REF_CKPT, map_location="cpu", weights_only=False)
    agent = _agent()
    agent._update_network  # noqa: B018  (no update needed: structure only)
    path = agent.save_checkpoint(str(tmp_path / "125_False1_model.pt"))
    ours = torch.load(path, map_location="cpu", weights_only=False)
    assert isinstance(ours, list) and len(ours) == len(ref) == 5
    for a, b in zip(ours[:4], ref[:4]):
        assert type(a) is type(b) and a.shape == b.shape and a.dtype == b.dtype
    assert list(ours[4].keys()) == list(ref[4].keys())
    for k in ref[4]:
        assert ours[4][k].shape == ref[4][k].shape and ours[4][k].dtype == ref[4][k].dtype
    # default file name: <save_dir>/<env_name>/<seed>_<add_demo><savetime>_model.pt (ddpg_agent.py:160-161)
    agent.args.save_dir = str(tmp_path)
    agent.model_path = os.path.join(agent.args.save_dir, agent.args.env_name)
    p2 = agent.save_checkpoint()
    assert os.path.basename(p2) == "125_False1_model.pt" and os.path.isfile(p2)


def test_reference_written_demo_file_preloads():
    """_init_demo_buffer (ddpg_agent.py:82-90) on a file written by the reference's get_push_demo."""
    demo = os.path.join(GOLDEN, "ref_written_6_push_demo.npz")
    agent = _agent(add_demo=True, demo_name=demo)
    d = np.load(demo, allow_pickle=True)
    n = d["obs"].shape[0]
    assert agent.buffer.current_size == n and agent.buffer.n_transitions_stored == 100 * n
    for key, src in (("obs", "obs"), ("ag", "ag"), ("g", "g"), ("actions", "acs")):
        assert np.array_equal(agent.buffer.buffers[key][:n], d[src]), key
    tr = agent.buffer.sample(32)
    assert tr["obs"].shape == (32, 27) and set(np.unique(tr["r"].view(np.uint32))) <= {0x80000000, 0xBF800000}
    assert agent.o_norm.total_count[0] == 1.0


@pytest.mark.parametrize("thr", [0.05, 0.1])
def test_compute_reward_and_is_success_ops_bit_exact(thr):
    """hp_compute_reward / hp_is_success vs outputs of the reference functions (bmirobot_env_push_F.py:84-90, 243-245)."""
    g = load_golden("reward_dense_success.npz")
    tag = f"thr{thr}"
    sparse = GoalDistanceReward(thr, "sparse", ctx=ctx())
    dense = GoalDistanceReward(thr, "dense", ctx=ctx())
    r = sparse.compute_reward(g["ag"], g["g"], None)
    assert r.dtype == np.float32 and r.shape == (g["ag"].shape[0],)
    assert np.array_equal(r.view(np.uint32), g[tag + "_sparse_bits"])
    rd = dense.compute_reward(g["ag"], g["g"], None)
    assert rd.dtype == np.float64 and np.array_equal(bits(rd), bits(g[tag + "_dense"]))
    ok = sparse._is_success(g["ag"], g["g"])
    assert ok.dtype == np.float32 and np.array_equal(bits(ok), bits(g[tag + "_success"]))
    # single pair (what env.step passes) and stacked leading dims (her.py:38 relies on the vectorisation)
    one = sparse.compute_reward(g["ag"][7], g["g"][7], None)
    assert np.asarray(one).shape == () and np.float32(one).view(np.uint32) == g[tag + "_sparse_bits"][7]
    if thr == 0.05:
        assert np.array_equal(bits(dense.compute_reward(g["stack_ag"], g["stack_g"], None)), bits(g["stack_dense"]))
        adv = load_golden("reward_adversarial.npz")
        assert np.array_equal(sparse.compute_reward(adv["ag"], adv["g"], None).view(np.uint32), adv["r_bits"])
    with pytest.raises(AssertionError):
        sparse.compute_reward(g["ag"][:4], g["g"][:5], None)          # goal_distance's shape assert (:21)


def test_reward_ops_on_device_arrays():
    g = load_golden("reward_dense_success.npz")
    n = g["ag"].shape[0]
    dev = torch.device("cuda", ctx().device_id)
    ag, gg = torch.from_numpy(g["ag"]).to(dev), torch.from_numpy(g["g"]).to(dev)
    out32 = torch.empty(n, dtype=torch.float32, device=dev)
    out64 = torch.empty(n, dtype=torch.float64, device=dev)
    ok = torch.empty(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    GoalDistanceReward(0.05, "sparse", ctx=ctx()).compute_reward_device(ag.data_ptr(), gg.data_ptr(), n, 3, out32.data_ptr())
    GoalDistanceReward(0.05, "dense", ctx=ctx()).compute_reward_device(ag.data_ptr(), gg.data_ptr(), n, 3, out64.data_ptr())
    GoalDistanceReward(0.05, "sparse", ctx=ctx()).is_success_device(ag.data_ptr(), gg.data_ptr(), n, 3, ok.data_ptr())
    ctx().synchronize()
    assert np.array_equal(out32.cpu().numpy().view(np.uint32), g["thr0.05_sparse_bits"])
    assert np.array_equal(bits(out64.cpu().numpy()), bits(g["thr0.05_dense"]))
    assert np.array_equal(bits(ok.cpu().numpy()), bits(g["thr0.05_success"]))


def test_dense_dict_sampler_returns_the_envs_float64_reward_from_the_device():
    """replay_buffer.sample with a dense-reward env: r is compute_reward's float64 -d (:89-90), computed by the gather
    kernel (no host arithmetic), bit-identical to the oracle's."""
    n, B, k, seed = 12, 200, 4, 31
    eps = make_episodes(n, seed=17, mode="walk")
    rng = fresh_rng(seed)
    sampler = her_sampler("future", k, GoalDistanceReward(0.05, "dense", ctx=ctx()).compute_reward, rng=rng)
    assert sampler.reward_type == "dense" and sampler.sq_threshold < 0
    buf = replay_buffer(dict(ENV_PARAMS), n * 100, sampler.sample_her_transitions, rng=rng)
    buf.store_episode(eps)
    tr = buf.sample(B)
    rs = np.random.RandomState(seed)
    st = EpisodeStore(100, 27, 3, 4, n * 100)
    st.store_episode(eps, rs)
    want, _ = st.sample(B, future_probability("future", k), rs,
                        reward_fn=lambda a, g: compute_reward(a, g, 0.05, "dense"))
    assert tr["r"].dtype == np.float64 and tr["r"].shape == (B, 1)
    for kk in want:
        assert np.array_equal(bits(tr[kk]), bits(want[kk])), kk
