"""Where the driver's 20-step timed region spends its time beyond 20 x the steady-state step: enqueue, graph launch, syncs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import argparse
import torch
import bench
a = argparse.Namespace(gpus=1, steps=20, warmup=5, batch=256, episodes=5000, replay_k=4, feeder_episodes=0, feeder_envs=0,
                       feeder_workers=8, no_cpu_baseline=True, no_profile=True, cpu_seconds=1.0)
r = bench.Runner(a, 0, 1)
r.run_steps(40); r.sync()
r.run_steps(5); r.run_steps(20); r.run_steps(15); r.sync()
r.run_steps(5); r.sync()
for rep in range(5):
    t0 = time.perf_counter()
    r.run_steps(20)
    t1 = time.perf_counter()
    r.ctx.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"enqueue {1e6*(t1-t0):.1f} us | ctx.synchronize {1e6*(t2-t1):.1f} | torch.cuda.synchronize {1e6*(t3-t2):.1f} | total {1e6*(t3-t0):.1f} = {1e6*(t3-t0)/20:.2f} us/step")
    r.run_steps(20); r.sync()     # back to the same cycle position (5 + 20 + 20 -> 45 = 5 into the next cycle)
    r.run_steps(35); r.run_steps(5); r.sync()
