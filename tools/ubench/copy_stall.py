"""Which runtime call owns the one-off ~30 ms stall seen at about the 150th episode store of a process?"""
import time, torch
dev = torch.device("cuda", 0)
s = torch.cuda.Stream()
src = torch.empty(120 * 1024 // 8, dtype=torch.float64).pin_memory()
dst = torch.empty_like(src, device=dev)
busy = torch.empty(1 << 20, device=dev)

def run(name, fn, n=600):
    torch.cuda.synchronize()
    worst = []
    for i in range(n):
        t0 = time.perf_counter()
        fn(i)
        dt = time.perf_counter() - t0
        if dt > 2e-3:
            worst.append((i, round(1e3 * dt, 2)))
    torch.cuda.synchronize()
    print(name, "iterations slower than 2 ms:", worst)

evs = [torch.cuda.Event() for _ in range(4)]
with torch.cuda.stream(s):
    def copy_only(i):
        dst.copy_(src, non_blocking=True); busy.add_(1.0)
    def copy_event(i):
        dst.copy_(src, non_blocking=True); e = evs[i % 4]; e.record(s); busy.add_(1.0); e.synchronize()
    def event_only(i):
        busy.add_(1.0); e = evs[i % 4]; e.record(s); e.synchronize()
    def copy_sizes(i):
        n = 1024 * (1 + (i % 7)); dst[:n].copy_(src[:n], non_blocking=True); busy.add_(1.0)
    run("event_only", event_only)
    run("copy_only", copy_only)
    run("copy_event", copy_event)
    run("copy_small_sizes", copy_sizes)
