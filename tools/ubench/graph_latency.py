"""Fixed cost of one hp_agent_sample_and_update(n) call (cached hipGraph of n updates) on an idle GPU: wall time from the call
to the end of ctx.synchronize(), minus n x the steady-state update.  How much of the driver's 20-step region is launch latency,
and does it grow with the number of graph nodes?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import argparse
import numpy as np
import bench
a = argparse.Namespace(gpus=1, steps=20, warmup=5, batch=256, episodes=5000, replay_k=4, feeder_episodes=0, feeder_envs=0,
                       feeder_workers=8, no_cpu_baseline=True, no_profile=True, cpu_seconds=1.0)
r = bench.Runner(a, 0, 1)
ag = r.agent
import torch
ts_stream = torch.cuda.Stream()
r.ctx.set_stream(ts_stream.cuda_stream)         # the library's launches and torch's events on one stream
r.run_steps(40); r.sync()
# steady state: 400 updates back to back
ag._update_network(40); r.sync()
t0 = time.perf_counter()
for _ in range(10): ag._update_network(40)
r.sync()
steady = 1e6 * (time.perf_counter() - t0) / 400
print(f"steady state {steady:.2f} us/update")
for n in (1, 2, 3, 5, 10, 20, 40):
    ag._update_network(n); r.sync()          # graph for this n is cached now
    ts = []
    evs = []
    for rep in range(9):
        time.sleep(0.002)                    # idle GPU, like a caller that waited for the previous result
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(ts_stream)
        ag._update_network(n)
        e1.record(ts_stream)
        t1 = time.perf_counter()
        r.ctx.synchronize()
        t2 = time.perf_counter()
        ts.append((1e6 * (t1 - t0), 1e6 * (t2 - t0)))
        evs.append(1e3 * e0.elapsed_time(e1))
    enq = float(np.median([x[0] for x in ts])); tot = float(np.median([x[1] for x in ts]))
    ev = float(np.median(evs))
    print(f"n={n:3d}: call returns after {enq:6.1f} us, done after {tot:7.1f} us -> fixed cost {tot - n * steady:6.1f} us ({(tot - n * steady) / n:5.2f} us/update); "
          f"event pair around the call on the stream: {ev:7.1f} us = {ev / n:5.2f} us/update on the GPU; host-only part {tot - ev:5.1f} us")

# a reference-style inner loop: one _update_network() call per minibatch (ddpg_agent.py:145-147), calls issued back to back
for n in (1, 2, 4):
    ag._update_network(n); r.sync()
    t0 = time.perf_counter()
    for _ in range(400 // n): ag._update_network(n)
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    print(f"loop of _update_network({n}) calls: {1e6 * (t2 - t0) / 400:.2f} us/update (host issues a call every {1e6 * (t1 - t0) / (400 // n):.1f} us)")

# the reference's own call form: argument-less calls are counted by the mirror and issued together (_lib.py, deferred updates)
ag._update_network(40); r.sync()
t0 = time.perf_counter()
for _ in range(400): ag._update_network()
r.sync()
print(f"loop of argument-less _update_network() calls (deferred, issued 40 at a time): {1e6 * (time.perf_counter() - t0) / 400:.2f} us/update")

# the reference's six lines of learn() (ddpg_agent.py:143-150) exactly as they stand, against train_cycle on the same episodes
from rl_arm_under_sparse_reward_amd.synthetic import make_episodes
eps = make_episodes(2, seed=5)
def six_lines():
    ag.buffer.store_episode(eps)
    ag._update_normalizer(eps)
    for _ in range(ag.args.n_batches):
        ag._update_network()
    ag._soft_update_target_network(ag.actor_target_network, ag.actor_network)
    ag._soft_update_target_network(ag.critic_target_network, ag.critic_network)
for name, fn in (("the reference's six lines, unchanged", six_lines), ("train_cycle", lambda: ag.train_cycle(eps))):
    for _ in range(3): fn()
    r.sync()
    t0 = time.perf_counter()
    for _ in range(50): fn()
    r.sync()
    dt = time.perf_counter() - t0
    print(f"{name}: {1e6 * dt / 50:.0f} us per cycle = {1e6 * dt / (50 * ag.args.n_batches):.2f} us per update")
