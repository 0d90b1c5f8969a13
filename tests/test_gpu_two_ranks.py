"""Two and FOUR data-parallel ranks on the GPU box (all processes share cuda:0; gloo process group; collectives through the
library's peer-memory exchange -- one-shot at 2 ranks, the reduce-scatter + all-gather form that is the DEFAULT from 4 ranks --
or through torch.distributed on host-staged copies of the library's device vectors).  Each rank samples its own minibatches from
its own shard with its own stream (seed + rank, train.py:36); gradients are SUMMED between backward and Adam
(utils.py:43-48), the normalizer's local sums are AVERAGED (normalizer.py:60-64), parameters start from rank 0's
(utils.py:6-15).  Checked per rank against the oracle run as two ranks over the same process group, and across ranks
for bit-identical networks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu
N_UP = 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, transport="torch"):
    import sys
    split = transport.endswith("split")
    if split:
        # Round 6: data-parallel ranks on a device each take the split launch (slab8_split.h) like a single rank does: the critic's
        # tiles exchange INSIDE the chain launch (k_fb_split8<1>), the actor's in k_gemm_lds_adam_peer behind it.  Ranks that
        # share a device (gate kernels on) keep the two-launch form -- the split launch's in-launch waits need the whole launch
        # resident, which eight ranks on one device do not give it -- so the rehearsal runs like "peertiles": gates off, batch 64
        # (2 x 48 chains: every chain of both ranks resident), split forced for the 8-update sequences of this test.
        transport = transport[:-5]
        os.environ["RLARM_SPLIT"] = "1"
    if transport == "auto":                  # nothing forced: the library picks transport and form (peer memory; two-phase from 4 ranks)
        os.environ.pop("RLARM_COMM", None)
        os.environ.pop("RLARM_PEER_PHASES", None)
    else:
        os.environ["RLARM_COMM"] = "peer" if transport == "peer2" else transport
    if transport == "peer":                  # the one-shot form, also where the default would be two-phase (4 ranks)
        os.environ["RLARM_PEER_PHASES"] = "1"
    if transport == "peer2":                 # reduce-scatter + all-gather over the same peer memory (default from 4 ranks)
        os.environ["RLARM_PEER_PHASES"] = "2"
    if transport.startswith("peertiles"):
        # Round 4: the tile-wise exchange INSIDE the weight-gradient launch (gemm_lds.h PEER), which ranks that share a device
        # normally avoid (their waits go into gate kernels and the exchange stays a launch of its own).  Forced here with the
        # gates off and a batch small enough that both ranks' launches are co-resident on the one device (53 chain workgroups
        # beside the other rank's waiting tiles), so that the flag rows, the rank-ordered sums and the buffer ping-pong of the
        # form every real multi-GPU job runs are exercised by two real processes.  Waits are bounded: a starved launch fails
        # the test after 5 s instead of hanging the box.
        os.environ["RLARM_COMM"] = "peer"
        os.environ["RLARM_PEER_PHASES"] = "1"
        os.environ["RLARM_PEER_TIMEOUT_S"] = "5"
    if transport == "peertilesks":
        # with a SPLIT reduction of the narrow problems (default from batch 768) the workgroup that reaches the
        # exchange is whichever slice arrived last -- a different workgroup index on every rank -- so the flag row must name the
        # tile, not the workgroup.  Forced at a batch that still lets both ranks' launches share the device.
        os.environ["RLARM_DW_KSPLIT"] = "2"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ddpg_update as oupd
    from oracle.her_replay import EpisodeStore, future_probability
    from oracle.running_norm import RunningNorm, update_normalizers
    from rl_arm_under_sparse_reward_amd import _lib
    from rl_arm_under_sparse_reward_amd.arguments import Args
    from rl_arm_under_sparse_reward_amd.ddpg_agent import NET_ACTOR, NET_CRITIC, ddpg_agent
    from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
    from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes
    from rl_arm_under_sparse_reward_amd.utils import Communicator

    torch.set_num_threads(2)
    comm = Communicator(0, gate=False if transport.startswith("peertiles") else None)
    assert comm.active and comm.world_size == world
    n_eps, batch, seed = 32, {"peertiles": 64, "peertilesks": 128}.get(transport, 256), 125 + rank
    eps = make_episodes(n_eps, seed=40 + rank, mode="walk")
    # ---- device side
    torch.manual_seed(100 + rank)            # ranks start from DIFFERENT nets; sync_networks must fix that (C1)
    rng = DeviceRandomState(seed)
    agent = ddpg_agent(Args(batch_size=batch, buffer_size=n_eps * 100), None, dict(ENV_PARAMS), comm=comm, rng=rng)
    assert agent._native_comm is None        # gloo group: no RCCL (two ranks share one device)
    peer = transport.startswith("peer") or transport == "auto"
    assert (agent._peer is not None) == peer
    if peer:
        import ctypes as C
        ph = C.c_int32()
        _lib.check(agent.lib.hp_peer_phases(agent._peer, C.byref(ph)))
        assert ph.value == (2 if transport == "peer2" or (transport == "auto" and world >= 4) else 1)
    a0 = {k: v.detach().clone() for k, v in agent.actor_network.state_dict().items()}
    c0 = {k: v.detach().clone() for k, v in agent.critic_network.state_dict().items()}
    agent.buffer.store_episode(eps)
    agent._update_normalizer()               # on the staged episodes; MEAN over ranks inside
    agent._update_network(N_UP)
    got = agent.last_losses(N_UP)
    # what one update launched, read off the library's launch logic (a host-driven exchange reports ONE update).  Asked AFTER the
    # first sequence: asked before it, the discarded capture loads the kernels' code early and both ranks then reach their first
    # tile-wise launch in the same microseconds -- the one timing the shared-device rehearsal of that form cannot survive (its
    # launches must overlap with a skew, see the fixture); measured 2 of 3 attempts lost that way, none this way
    kernels = agent.update_kernels(N_UP)["updates"][-1]
    if split:
        assert kernels[0] == "k_fb_split8<1>", kernels
    elif transport != "torch":
        assert kernels[0] == "k_fb_slab8", kernels
    # ---- oracle side, same collectives in the same order on both ranks
    def ar_sum(x):
        # MPI_SUM in rank order 0..W-1 in the array's own dtype (what tools/gen_golden.py's stub communicator does for the
        # multi-rank fixtures, and what the device sums): from 3 ranks on a float32 sum depends on the order, and gloo's
        # all_reduce picks its own
        t = torch.from_numpy(np.array(x, copy=True))
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        acc = every[0].numpy().copy()
        for e in every[1:]:
            acc = acc + e.numpy()
        return acc
    def ar_mean(x):
        return ar_sum(x) / world
    rs = np.random.RandomState(seed)
    st = EpisodeStore(100, 27, 3, 4, n_eps * 100)
    st.store_episode(eps, rs)
    fp = future_probability("future", 4)
    on = RunningNorm(27, default_clip_range=5, allreduce_mean=ar_mean)
    gn = RunningNorm(3, default_clip_range=5, allreduce_mean=ar_mean)
    update_normalizers(on, gn, eps, fp, rs)
    learner = oupd.DDPGLearner(a0, c0, allreduce_sum=ar_sum)
    want = []
    for _ in range(N_UP):
        tr, _ = st.sample(batch, fp, rs)
        res = learner.update(*oupd.minibatch_tensors(tr, on, gn))
        want.append([res["actor_loss"], res["critic_loss"]])
    out = {"got": got, "want": np.array(want), "kernels": kernels, "actor": agent._get_flat(NET_ACTOR), "critic": agent._get_flat(NET_CRITIC),
           "actor0": oupd.flatten(list(a0.values())), "oracle_actor": learner.flat("actor"),
           "oracle_critic": learner.flat("critic"), "critic0": oupd.flatten(list(c0.values())),
           "rng_equal": bool(np.array_equal(rng.get_state()[1], rs.get_state()[1]) and rng.get_state()[2] == rs.get_state()[2]),
           "o_mean": np.asarray(agent.o_norm.mean), "oracle_o_mean": np.asarray(on.mean)}
    if peer:
        # normalizer._mpi_average (normalizer.py:60-64) through the peer mailboxes, on stand-alone normalizers fed the golden
        # recipe: the rank-ordered sum / world must give the reference's bits (tests/golden/normalizer.npz w{world}_*, the
        # reference run on `world` stub ranks)
        from conftest import load_golden
        from rl_arm_under_sparse_reward_amd.normalizer import normalizer as dev_normalizer
        g = load_golden("normalizer.npz")

        def golden_inputs(step, size):
            rs_ = np.random.RandomState(1000 + 17 * rank + step)
            n_ = [100, 100, 37, 250, 1, 100][step % 6]
            scale = [1.0, 30.0, 1e-3, 250.0, 1.0, 5.0][step % 6]
            return rs_.normal(0.3 * (rank + 1), scale, size=(n_, size))

        golden_ok = True
        for size in (27, 3):
            nz = dev_normalizer(size, default_clip_range=5, std_dtype=str(g["std_dtype"]), comm=comm)
            i = 0
            for step in range(6):
                nz.update(np.clip(golden_inputs(step, size), -200, 200))
                if step % 2 == 1 or step == 4:
                    nz.recompute_stats()
                    for nm in ("mean", "std", "total_sum", "total_sumsq", "total_count"):
                        ref = g[f"w{world}_d{size}_r{i}_{nm}"]
                        got = np.asarray(getattr(nz, nm))
                        golden_ok = golden_ok and got.dtype == ref.dtype and np.array_equal(got.view(np.uint8), ref.view(np.uint8))
                    i += 1
        out["normalizer_golden_ok"] = bool(golden_ok)
        # the whole cycle as ONE hipGraph with the exchange inside (gradients per update, normalizer sums once)
        import ctypes as C
        more = make_episodes(2, seed=70 + rank, mode="walk")
        agent.train_cycle(more, n_batches=5)
        agent.train_cycle(make_episodes(2, seed=80 + rank, mode="walk"), n_batches=5)
        mode, err = C.c_int32(), C.c_uint32()
        _lib.check(agent.lib.hp_agent_cycle_mode(agent.h, C.byref(mode)))
        _lib.check(agent.lib.hp_peer_status(agent._peer, C.byref(err)))
        out.update(cycle_mode=mode.value, peer_error=err.value, actor_after_cycles=agent._get_flat(NET_ACTOR),
                   g_std_after_cycles=np.asarray(agent.g_norm.std), losses_after=agent.last_losses(10))
    torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    _lib.Context.default().synchronize()
    agent.close_comm()
    dist.destroy_process_group()


@pytest.fixture(scope="module", params=[("torch", 2), ("peer", 2), ("peer2", 2), ("peertiles", 2), ("peertilesks", 2), ("auto", 4), ("torch", 4), ("peer", 4),
                                        ("peertilessplit", 2)],
                ids=lambda p: f"{p[0]}-w{p[1]}")
def two_ranks(request, tmp_path_factory):
    """torch: collectives through torch.distributed (gloo, host-staged) from a host-driven loop.
    peer: the library's one-shot all-reduce over IPC-mapped peer memory, fused with Adam (csrc/peer.hip) -- the two
    processes map each other's exchange block on the shared device, which exercises flags, epochs, buffer ping-pong and the
    rank-ordered sum exactly as two GPUs would (the fabric itself only exists on a multi-GPU node).
    peer2: the same memory used as reduce-scatter + all-gather (each rank sums its slice, a second flag round, every rank
    gathers the reduced slices), the default from 4 ranks.
    auto (4 ranks): nothing forced -- what a 4-GPU job gets: peer memory, two-phase form; (peer, 4) forces the one-shot form
    at 4 ranks, (torch, 4) the host-driven fallback.  All against the 4-rank oracle."""
    transport, world = request.param
    out = tmp_path_factory.mktemp(f"gpu{world}_{transport}")
    if transport.startswith("peertiles"):
        # The tile-wise form lets a launch WAIT for the peer's launch: on real ranks (a device each) that is the design; with
        # two processes on ONE device it only works while both launches are resident together, which the dispatcher does not
        # promise (one rank's waiting tiles can hold the LDS the other rank's chain workgroups need).  A rehearsal that starved
        # fails its bounded waits loudly after 5 s -- tried up to three times, then skipped as "not co-resident", never
        # passed silently.  (The same kernels run bitwise against the single-rank path at world 1: test_gpu_update.py
        # `+peer` variants, test_gpu_teacher_forced.py.)
        for attempt in range(3):
            try:
                mp.spawn(_worker, args=(world, _free_port(), str(out), transport), nprocs=world, join=True)
                break
            except Exception as e:      # mp.spawn re-raises the first failing rank's error as ProcessRaisedException
                if not any(t in str(e) for t in ("exchange is dead", "waited longer than the bound", "timed out")):
                    raise
                print(f"[peertiles rehearsal] attempt {attempt + 1}: the two ranks' launches were not co-resident (bounded waits gave up)")
                if attempt == 2:
                    pytest.skip("two ranks' launches were not co-resident on the one device in three attempts: " + str(e)[-200:])
    else:
        mp.spawn(_worker, args=(world, _free_port(), str(out), transport), nprocs=world, join=True)
    res = [torch.load(os.path.join(out, f"rank{r}.pt"), weights_only=False) for r in range(world)]
    for r in res:
        r["transport"] = transport
    return res


def test_ranks_end_with_identical_networks(two_ranks):
    r0 = two_ranks[0]
    for r1 in two_ranks[1:]:
        assert np.array_equal(r0["actor0"], r1["actor0"]) and np.array_equal(r0["critic0"], r1["critic0"])   # C1
        assert np.array_equal(r0["actor"].view(np.uint8), r1["actor"].view(np.uint8))       # same summed gradients, same Adam
        assert np.array_equal(r0["critic"].view(np.uint8), r1["critic"].view(np.uint8))
        assert np.array_equal(r0["o_mean"].view(np.uint8), r1["o_mean"].view(np.uint8))     # C4


def test_each_rank_tracks_the_two_rank_oracle(two_ranks):
    for r in two_ranks:
        assert r["rng_equal"]                                   # the sampler consumed exactly the oracle's words
        if r["transport"] == "torch" and len(two_ranks) > 2:
            # a collective library sums in its own order (gloo here, RCCL on the fabric): from 3 ranks on that is a last-bit
            # matter; the peer exchange sums in rank order like the oracle and is held to the bits
            assert np.allclose(r["o_mean"], r["oracle_o_mean"], rtol=1e-6, atol=1e-9)
        else:
            assert np.array_equal(r["o_mean"].view(np.uint8), r["oracle_o_mean"].view(np.uint8))
        for i in range(N_UP):                                   # first update: the north-star 1e-5; chained ones: 1e-4
            tol = 1e-5 if i == 0 else 1e-4
            for j in range(2):
                assert abs(r["got"][i, j] - r["want"][i, j]) <= tol * max(abs(r["want"][i, j]), 1e-2), (i, j, r["got"][i], r["want"][i])
        for name in ("actor", "critic"):
            moved = np.linalg.norm(r[f"oracle_{name}"] - r[f"{name}0"])
            assert np.linalg.norm(r[name] - r[f"oracle_{name}"]) <= 0.05 * moved


def test_peer_exchange_keeps_ranks_identical_through_graph_cycles(two_ranks):
    r0 = two_ranks[0]
    if r0["transport"] == "torch":
        pytest.skip("peer-memory transport only")
    print(f"[{r0['transport']} x {len(two_ranks)}] kernels of one update: {r0['kernels']}")
    for r1 in two_ranks[1:]:
        assert r0["normalizer_golden_ok"] and r1["normalizer_golden_ok"]   # _mpi_average through the mailboxes: reference bits
        assert r0["cycle_mode"] == 1 and r1["cycle_mode"] == 1          # the cycle, exchange included, replays as a hipGraph
        assert r0["peer_error"] == 0 and r1["peer_error"] == 0
        assert np.array_equal(r0["actor_after_cycles"].view(np.uint8), r1["actor_after_cycles"].view(np.uint8))
        assert np.array_equal(r0["g_std_after_cycles"].view(np.uint8), r1["g_std_after_cycles"].view(np.uint8))
        assert np.all(np.isfinite(r0["losses_after"])) and not np.array_equal(r0["losses_after"], r1["losses_after"])


def _late_worker(rank, world, port, out_dir):
    """rank 1 arrives 4 s late at an update whose wait bound is 1 s."""
    import sys
    import time
    os.environ["RLARM_COMM"] = "peer"
    os.environ["RLARM_PEER_TIMEOUT_S"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rl_arm_under_sparse_reward_amd import _lib
    from rl_arm_under_sparse_reward_amd.arguments import Args
    from rl_arm_under_sparse_reward_amd.ddpg_agent import NET_ACTOR, ddpg_agent
    from rl_arm_under_sparse_reward_amd.random import DeviceRandomState
    from rl_arm_under_sparse_reward_amd.synthetic import ENV_PARAMS, make_episodes
    from rl_arm_under_sparse_reward_amd.utils import Communicator

    comm = Communicator(0)
    torch.manual_seed(0)
    agent = ddpg_agent(Args(batch_size=64, buffer_size=800), None, dict(ENV_PARAMS), comm=comm,
                       rng=DeviceRandomState(5 + rank))
    assert agent._peer is not None
    agent.buffer.store_episode(make_episodes(8, seed=3 + rank, mode="walk"))
    agent._update_normalizer()
    agent._update_network(2)                     # in step: fine
    agent.ctx.synchronize()
    agent.check_exchange()
    before = agent._get_flat(NET_ACTOR)
    dist.barrier()
    if rank == 1:
        time.sleep(4.0)                          # a rollout / checkpoint that outlasts the bound
    t0 = time.time()
    agent._update_network(1)
    agent.ctx.synchronize()
    waited = time.time() - t0
    raised_status = raised_call = False
    try:
        agent.check_exchange()
    except RuntimeError:
        raised_status = True
    try:
        agent._update_network(1)                 # the next call fails without any synchronisation (pinned error word)
        agent.ctx.synchronize()
        agent._update_network(1)
    except _lib.HpError as e:
        raised_call = "exchange is dead" in str(e)
    t1 = time.time()
    try:
        agent.train_cycle(make_episodes(2, seed=9, mode="walk"), 2)
    except _lib.HpError:
        pass
    out = {"waited": waited, "raised_status": raised_status, "raised_call": raised_call, "second_call_s": time.time() - t1,
           "stepped": not np.array_equal(before, agent._get_flat(NET_ACTOR))}
    torch.save(out, os.path.join(out_dir, f"late{rank}.pt"))
    dist.barrier()
    agent.close_comm()
    dist.destroy_process_group()


def test_a_rank_late_beyond_the_wait_bound_is_fatal_not_silent(tmp_path):
    """a timed-out wait used to set a word nobody read while the kernel summed whatever the peers' buffers held.
    Now the kernel whose wait gives up skips its optimizer step, later exchange kernels return at once, and the next host
    call raises on the rank that waited; the late rank finds its peer gone and fails the same way one update later."""
    mp.spawn(_late_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"late{r}.pt"), weights_only=False) for r in range(2))
    assert 0.9 <= r0["waited"] <= 3.5                  # rank 0 gave up after the 1 s bound, not after 20 s and not at once
    assert r0["raised_status"] and r0["raised_call"]
    assert not r0["stepped"]                           # no Adam step from a partial sum
    assert r0["second_call_s"] < 1.0                   # dead exchange: no further stall per call
    assert r1["raised_call"] or r1["raised_status"]    # the late rank fails too (its peer stopped signalling)
