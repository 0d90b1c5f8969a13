"""W ranks sharing cuda:0 (gloo group): attach the peer exchange, run sampled updates, print per-rank time and exchange
status.  python tools/ubench/ranks_probe.py WORLD [n_updates] [cycles]   (env: RLARM_PEER_PHASES, RLARM_PEER_TIMEOUT_S, ...)"""
import os
import socket
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, n_up, cycles):
    import ctypes as C

    import torch
    import torch.distributed as dist
    sys.path.insert(0, REPO)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rl_arm_under_sparse_reward_amd import _lib
    from rl_arm_under_sparse_reward_amd.arguments import Args
    from rl_arm_under_sparse_reward_amd.ddpg_agent import ddpg_agent
    from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
    from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes
    from rl_arm_under_sparse_reward_amd.utils import Communicator
    comm = Communicator(0)
    torch.manual_seed(0)
    t0 = time.time()
    agent = ddpg_agent(Args(batch_size=256, buffer_size=6400), None, dict(ENV_PARAMS), comm=comm, rng=DeviceRandomState(5 + rank))
    print(f"[{rank}] attach {time.time() - t0:.2f}s peer={agent._peer is not None}", flush=True)
    agent.buffer.store_episode(make_episodes(32, seed=3 + rank, mode="walk"))
    agent._update_normalizer()
    agent.ctx.synchronize()

    def status():
        err = C.c_uint32()
        if agent._peer is None:
            return -1
        agent.lib.hp_peer_status(agent._peer, C.byref(err))
        return err.value
    print(f"[{rank}] normalizer ok status {status():#x}", flush=True)
    for n in (1, 2, n_up):
        dist.barrier()
        t0 = time.time()
        try:
            agent._update_network(n)
            agent.ctx.synchronize()
        except Exception as e:
            print(f"[{rank}] update({n}) raised {e}", flush=True)
            break
        print(f"[{rank}] update({n}) {1e3 * (time.time() - t0):.1f} ms status {status():#x}", flush=True)
    for c in range(cycles):
        dist.barrier()
        t0 = time.time()
        try:
            agent.train_cycle(make_episodes(2, seed=50 + c + rank, mode="walk"), n_up)
            agent.ctx.synchronize()
        except Exception as e:
            print(f"[{rank}] cycle {c} raised {e}", flush=True)
            break
        print(f"[{rank}] cycle {c} {1e3 * (time.time() - t0):.1f} ms status {status():#x}", flush=True)
    dist.barrier()
    agent.close_comm()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1])
    n_up = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(world, port, n_up, cycles), nprocs=world, join=True)
