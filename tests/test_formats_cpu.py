"""On-disk formats either side of the hot path, pinned by files the REFERENCE ITSELF wrote (tools/gen_golden.py):
  tests/golden/ref_checkpoint_model.pt      the torch.save statement of ddpg_agent.py:158-161 executed on a reference agent
  tests/golden/ref_checkpoint_probe.npz     what the reference's reader (demo_push.py:15-22 + models.actor) computes from it
  tests/golden/ref_written_6_push_demo.npz  get_push_demo (get_demo_data_push.py:24-94) driving a stand-in GoalEnv
  tests/golden/reward_dense_success.npz     compute_reward 'dense' / _is_success (bmirobot_env_push_F.py:84-90, 243-245)
CPU half: the oracle reads these files like the reference does; our writers produce the same schema; and (build
container only) a checkpoint in our payload layout loads into the reference's own models.actor."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bits, load_golden
from oracle import ddpg_update as oupd
from oracle.her_replay import compute_reward, is_success
from oracle.running_norm import RunningNorm

REF = "/root/reference"
KEYS = ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias", "action_out.weight",
        "action_out.bias"]
ENV_PARAMS = {"obs": 27, "goal": 3, "action": 4, "action_max": 0.5, "max_timesteps": 100}


def _load_ref_checkpoint():
    return torch.load(os.path.join(GOLDEN, "ref_checkpoint_model.pt"), map_location="cpu", weights_only=False)


def test_reference_checkpoint_structure():
    o_mean, o_std, g_mean, g_std, model = _load_ref_checkpoint()
    pr = load_golden("ref_checkpoint_probe.npz")
    assert list(model.keys()) == KEYS == list(pr["keys"])
    assert model["fc1.weight"].shape == (256, 30) and model["action_out.weight"].shape == (4, 256)
    assert all(v.dtype == torch.float32 for v in model.values())
    for a, n in ((o_mean, 27), (o_std, 27), (g_mean, 3), (g_std, 3)):
        assert isinstance(a, np.ndarray) and a.shape == (n,)
    assert o_mean.dtype == np.float32 and g_mean.dtype == np.float32
    for nm, a in (("o_mean", o_mean), ("o_std", o_std), ("g_mean", g_mean), ("g_std", g_std)):
        assert np.array_equal(bits(a), bits(pr[nm])), nm


def test_oracle_reads_reference_checkpoint_like_demo_push():
    """oracle = RunningNorm.normalize on clipped inputs + actor_forward; must reproduce the reference reader's outputs."""
    o_mean, o_std, g_mean, g_std, model = _load_ref_checkpoint()
    pr = load_golden("ref_checkpoint_probe.npz")
    on, gn = RunningNorm(27, default_clip_range=float(pr["clip_range"])), RunningNorm(3, default_clip_range=float(pr["clip_range"]))
    on.mean, on.std, gn.mean, gn.std = o_mean, o_std, g_mean, g_std
    co = float(pr["clip_obs"])
    x = np.concatenate([on.normalize(np.clip(pr["probe_obs"], -co, co)), gn.normalize(np.clip(pr["probe_g"], -co, co))], axis=1)
    xt = torch.tensor(x, dtype=torch.float32)
    assert np.array_equal(xt.numpy(), pr["inputs"])                 # bit-identical network inputs
    got = oupd.actor_forward(model, xt, 0.5).numpy()
    assert np.allclose(got, pr["actions"], rtol=1e-6, atol=1e-7)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_our_checkpoint_layout_loads_into_the_reference_actor(tmp_path):
    """A checkpoint in the layout ddpg_agent.checkpoint_payload() produces -- [mean, std, mean, std, state_dict of our
    models.actor container] -- goes through the reference's reader: torch.load + models.actor.load_state_dict (strict)
    + process_inputs (demo_push.py:15-22,28,41)."""
    from rl_arm_under_sparse_reward_amd.models import actor as our_actor

    torch.manual_seed(3)
    ours = our_actor(dict(ENV_PARAMS))               # unattached container: state_dict() is the host tensors
    rs = np.random.RandomState(0)
    payload = [rs.normal(size=27).astype(np.float32), np.abs(rs.normal(size=27)) + 0.1,
               rs.normal(size=3).astype(np.float32), np.abs(rs.normal(size=3)) + 0.1, ours.state_dict()]
    path = str(tmp_path / "125_False1_model.pt")
    torch.save(payload, path)
    sys.path.insert(0, REF)
    try:
        import models as ref_models
    finally:
        sys.path.remove(REF)
    o_mean, o_std, g_mean, g_std, model = torch.load(path, map_location=lambda storage, loc: storage, weights_only=False)
    net = ref_models.actor(dict(ENV_PARAMS))
    net.load_state_dict(model)                        # strict: key names and shapes must match the reference module
    net.eval()
    o, g = rs.uniform(-1, 1, 27), rs.uniform(0, 0.5, 3)
    x = np.concatenate([np.clip((np.clip(o, -200, 200) - o_mean) / o_std, -5, 5),
                        np.clip((np.clip(g, -200, 200) - g_mean) / g_std, -5, 5)])
    xt = torch.tensor(x, dtype=torch.float32)
    with torch.no_grad():
        a_ref = net(xt).numpy()
    a_orc = oupd.actor_forward(ours.state_dict(), xt[None], 0.5).numpy()[0]
    assert np.allclose(a_ref, a_orc, rtol=1e-6, atol=1e-7)


def test_reference_written_demo_schema_and_our_writer(tmp_path):
    from rl_arm_under_sparse_reward_amd.synthetic import write_demo_npz

    ref = np.load(os.path.join(GOLDEN, "ref_written_6_push_demo.npz"), allow_pickle=True)
    assert sorted(ref.files) == ["acs", "ag", "g", "info", "obs"]
    n = ref["obs"].shape[0]
    assert ref["obs"].shape == (n, 101, 27) and ref["ag"].shape == (n, 101, 3)
    assert ref["g"].shape == (n, 100, 3) and ref["acs"].shape == (n, 100, 4)
    assert ref["info"].shape == (n, 100) and ref["info"].dtype == object and "is_success" in ref["info"][0, 0]
    assert all(ref[k].dtype == np.float64 for k in ("obs", "ag", "g", "acs"))
    # only successful episodes are kept by the reference generator (:77)
    assert all(float(ref["info"][e, -1]["is_success"]) == 1.0 for e in range(n))
    ours_path = str(tmp_path / "bmirobot_3_push_demo.npz")
    write_demo_npz(ours_path, n_episodes=3, seed=1)
    ours = np.load(ours_path, allow_pickle=True)
    assert sorted(ours.files) == sorted(ref.files)
    for k in ref.files:
        assert ours[k].dtype == ref[k].dtype and ours[k].shape[1:] == ref[k].shape[1:], k
    assert set(ours["info"][0, 0]) == set(ref["info"][0, 0])


def test_oracle_dense_reward_and_success_vs_reference():
    g = load_golden("reward_dense_success.npz")
    for thr in (0.05, 0.1):
        tag = f"thr{thr}"
        rd = compute_reward(g["ag"], g["g"], thr, "dense")
        assert rd.dtype == np.float64 and np.array_equal(bits(rd), bits(g[tag + "_dense"]))
        sp = compute_reward(g["ag"], g["g"], thr, "sparse")
        assert np.array_equal(sp.view(np.uint32), g[tag + "_sparse_bits"])
        assert np.array_equal(bits(is_success(g["ag"], g["g"], thr)), bits(g[tag + "_success"]))
    assert np.array_equal(bits(compute_reward(g["stack_ag"], g["stack_g"], 0.05, "dense")), bits(g["stack_dense"]))
