"""Soak of the data-parallel split launch's in-launch exchange (k_fb_split8<1> + k_gemm_lds_adam_peer) with TWO real processes on the
one device (gates off, batch 64: every chain of both ranks resident; tests/test_gpu_two_ranks.py[peertilessplit-w2] is the short
form): CYCLES training cycles of UPDATES updates each, as cached hipGraphs -- thousands of exchange epochs through both buffer
parities and both flag-row ranges -- then: no wait timed out, the replicas' four networks bit-identical across ranks, losses finite
and different per rank (each rank samples its own shard).  Measurement / soak helper, not product code.
  CYCLES=300 UPDATES=12 python tools/ubench/dp_split_soak.py"""
import os
import socket
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, out_dir, cycles, updates):
    os.environ.update(RLARM_COMM="peer", RLARM_PEER_PHASES="1", RLARM_PEER_TIMEOUT_S="5", RLARM_SPLIT="1", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    from rl_arm_under_sparse_reward_amd import _lib
    from rl_arm_under_sparse_reward_amd.arguments import Args
    from rl_arm_under_sparse_reward_amd.ddpg_agent import ddpg_agent
    from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
    from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes
    from rl_arm_under_sparse_reward_amd.utils import Communicator

    comm = Communicator(0, gate=False)
    torch.manual_seed(100 + rank)
    agent = ddpg_agent(Args(batch_size=64, buffer_size=64 * 100), None, dict(ENV_PARAMS), comm=comm, rng=DeviceRandomState(125 + rank))
    assert agent._peer is not None
    agent.buffer.store_episode(make_episodes(32, seed=40 + rank, mode="walk"))
    pool = [make_episodes(2, seed=900 + 31 * rank + i, mode="walk") for i in range(8)]
    agent.train_cycle(pool[0], updates)                   # skewed start (graph capture), as in the test
    kernels = agent.update_kernels(updates)["updates"][-1]
    t0 = time.time()
    for c in range(cycles):
        agent.train_cycle(pool[c % len(pool)], updates)
        if c % 50 == 49:
            agent.ctx.synchronize()
            agent.check_exchange()
    agent.ctx.synchronize()
    dt = time.time() - t0
    agent.check_exchange()
    err = C.c_uint32()
    _lib.check(agent.lib.hp_peer_status(agent._peer, C.byref(err)))
    nets = [agent._get_flat(s) for s in (0, 1, 2, 3)]
    torch.save({"nets": nets, "losses": agent.last_losses(updates), "err": err.value, "kernels": kernels, "us_per_update": 1e6 * dt / (cycles * updates),
                "o_mean": np.asarray(agent.o_norm.mean)}, os.path.join(out_dir, f"soak{rank}.pt"))
    dist.barrier()
    agent.close_comm()
    dist.destroy_process_group()


def main():
    import tempfile
    cycles, updates = int(os.environ.get("CYCLES", "300")), int(os.environ.get("UPDATES", "12"))
    world = int(os.environ.get("WORLD", "2"))     # 3 / 4: every rank reads every peer's tile (3 x 48 and 4 x 48 chains still fit the CUs)
    for attempt in range(4):
        out = tempfile.mkdtemp()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        try:
            mp.spawn(worker, args=(world, port, out, cycles, updates), nprocs=world, join=True)
        except Exception as e:      # the shared-device rehearsal needs both launches co-resident: bounded waits give up loudly otherwise
            print(f"attempt {attempt + 1}: {str(e)[-300:]}")
            continue
        r = [torch.load(os.path.join(out, f"soak{k}.pt"), weights_only=False) for k in range(world)]
        same = all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for q in r[1:] for a, b in zip(r[0]["nets"], q["nets"]))
        print(f"world {world}: {cycles} cycles x {updates} updates = {cycles * updates} exchange epochs per rank; kernels {r[0]['kernels']}")
        print(f"replicas bit-identical (actor, critic, both targets): {same}; peer error words {r[0]['err']}, {r[1]['err']}; "
              f"normalizer means equal: {np.array_equal(r[0]['o_mean'], r[1]['o_mean'])}; losses finite: "
              f"{bool(np.all(np.isfinite(r[0]['losses'])) and np.all(np.isfinite(r[1]['losses'])))}, differ per rank: "
              f"{not np.array_equal(r[0]['losses'], r[1]['losses'])}; {r[0]['us_per_update']:.1f} us/update ({world} ranks on ONE device)")
        sys.exit(0 if same and all(q["err"] == 0 for q in r) else 1)
    sys.exit(2)


if __name__ == "__main__":
    main()
