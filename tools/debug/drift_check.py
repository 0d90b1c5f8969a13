"""Largest relative loss deviation from the torch-CPU oracle over 40 chained updates (what the tracking tests bound).
Test infrastructure (it runs the body of test_updates_track_oracle_over_a_cycle and reports instead of asserting):
    RLARM_ENGINE=slab32 python tools/debug/drift_check.py 3072"""
import os, sys
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_REPO, "tests"))
sys.path.insert(0, _REPO)
import numpy as np, torch
import test_gpu_update as T
batch = int(sys.argv[1])
src = open(T.__file__).read()
import re
# same body as the test, but report instead of assert
body = src[src.index("def test_updates_track_oracle_over_a_cycle"):src.index("@pytest.mark.parametrize(\"batch\", [64, 449, 1024, 4096])")]
body = body.replace("def test_updates_track_oracle_over_a_cycle(batch, k, engine=\"\", monkeypatch=None):", "def run(batch, k, engine=\"\", monkeypatch=None):")
body = re.sub(r"        assert abs\(got\[i, 0\].*\n", "        worst[0] = max(worst[0], abs(got[i, 0] - res['actor_loss']) / max(abs(res['actor_loss']), 1e-2)); trace.append(float(abs(got[i, 0] - res['actor_loss']) / max(abs(res['actor_loss']), 1e-2)))\n", body)
body = re.sub(r"        assert abs\(got\[i, 1\].*\n", "        worst[1] = max(worst[1], abs(got[i, 1] - res['critic_loss']) / max(abs(res['critic_loss']), 1e-2))\n", body)
worst = [0.0, 0.0]
trace = []
ns = dict(T.__dict__); ns["worst"] = worst; ns["trace"] = trace
exec(body, ns)
ns["run"](batch, int(os.environ.get("REPLAY_K", "4")))
if os.environ.get("TRACE"): print(" ".join(f"{v:.1e}" for v in trace))
print("batch", batch, os.environ.get("RLARM_DW64"), os.environ.get("RLARM_ENGINE"), "worst rel dev actor/critic", worst)
