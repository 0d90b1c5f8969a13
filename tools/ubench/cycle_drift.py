"""Per-segment timing of back-to-back training cycles (is the cycle path's rate stable over many cycles?)."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--segments", type=int, default=24); ap.add_argument("--cycles", type=int, default=10)
ap.add_argument("--mode", default="cycle")
x = ap.parse_args()
a = argparse.Namespace(gpus=1, steps=0, warmup=0, batch=256, episodes=5000, replay_k=4, feeder_episodes=0, feeder_envs=0)
r = bench.Runner(a, 0, 1)
import gc
if x.mode == "nogc":
    gc.collect(); gc.freeze(); gc.disable()
    x.mode = "cycle"
gc.callbacks.append(lambda phase, info: phase == "stop" and info["generation"] == 2 and print("gen2 collection", info, flush=True))
r.run_steps(40); r.sync()
out = []
for s in range(x.segments):
    t0 = time.perf_counter()
    if x.mode == "sleep" and s == 4:
        time.sleep(1.0)
        t0 = time.perf_counter()
    if x.mode == "updonly":     # graph replays only: no host copies, no boundary kernels
        for _ in range(x.cycles):
            r.agent._update_network(40)
    elif x.mode == "storeonly":  # the per-cycle host->device staging alone (plus 1 update to keep the stream busy)
        for _ in range(x.cycles * 8):
            r.agent.buffer.store_episode(r.pool[0]); r.agent._update_network(1)
    elif x.mode in ("cycle", "sleep"):
        r.run_steps(40 * x.cycles)
    else:                       # 39-step chunks: the eager/update-graph path with explicit boundaries
        for _ in range(x.cycles):
            r.run_steps(39); r.run_steps(1)
    t1 = time.perf_counter()
    r.sync()
    t2 = time.perf_counter()
    out.append((round(1e6 * (t2 - t0) / (40 * x.cycles), 2), round(1e6 * (t1 - t0) / (40 * x.cycles), 2)))
print(x.mode, "us/step per segment (total, host-enqueue part):", out)
