import os, sys
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_REPO, "tests")); sys.path.insert(0, _REPO)
import numpy as np, torch
import test_gpu_update as T
src = open(T.__file__).read()
body = src[src.index("def test_other_env_shapes_track_oracle"):src.index("@pytest.mark.parametrize(\"batch,want\"")]
body = body.replace("    assert np.allclose(agent.actor_network(x), want, rtol=1e-4, atol=2e-5)", "    d = np.abs(agent.actor_network(x) - want); print('max abs diff', d.max(), 'max |want|', np.abs(want).max(), 'losses', got[-1])")
ns = dict(T.__dict__); exec(body, ns)
ns["test_other_env_shapes_track_oracle"](60, 3, 7, 20)
