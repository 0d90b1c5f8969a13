"""Per-step cost of the data-parallel code path (forward_backward -> all-reduce -> apply) at world size 1
(forced collectives): isolates host/launch overhead of the N>1 path from the collective's wire time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from rl_arm_under_sparse_reward_amd import _lib
from rl_arm_under_sparse_reward_amd.arguments import Args
from rl_arm_under_sparse_reward_amd.ddpg_agent import ddpg_agent
from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes
from rl_arm_under_sparse_reward_amd.utils import Communicator
ctx = _lib.Context(0)
args = Args(batch_size=256, buffer_size=500000, replay_k=4, seed=125)
rng = DeviceRandomState(125, ctx=ctx)
torch.manual_seed(0)
ag = ddpg_agent(args, None, dict(ENV_PARAMS), comm=Communicator(0, force=True), ctx=ctx, rng=rng)
ag.buffer.store_episode(make_episodes(5000, seed=1)); ag._update_normalizer()
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
def steps(n):
    if mode == "cycle":
        for _ in range(n // 40): ag.train_cycle(make_eps, 40)
    else:
        ag._update_network(n)
make_eps = make_episodes(2, seed=3)
steps(200); ctx.synchronize(); torch.cuda.synchronize()
t0 = time.perf_counter(); steps(2000); ctx.synchronize(); torch.cuda.synchronize()
print(f"dp path ({mode}, RLARM_COMM={os.environ.get('RLARM_COMM', 'native')}, native={ag._native_comm is not None}): {1e6 * (time.perf_counter() - t0) / 2000:.1f} us/step")
import ctypes as C
m = C.c_int32(); _lib.check(ag.lib.hp_agent_cycle_mode(ag.h, C.byref(m))); print("cycle mode (0 none, 1 graph, 2 eager fallback):", m.value, "| last error:", ag.lib.hp_last_error())
dist.destroy_process_group()
