"""Scratch: the default bench line with every step of every config timed on its own (device synchronised after each)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def timed_steps(step, steps, warmup, barrier, reduce_max, after_warmup=None):
    times = []
    for i in range(warmup + steps):
        if i == warmup:
            if after_warmup is not None:
                after_warmup()
            barrier()
            t_all = time.perf_counter()
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        times.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    barrier()
    el = time.perf_counter() - t_all
    print("steps (enqueue ms, done ms):", " ".join(f"({a:.1f},{b:.1f})" for a, b in times), file=sys.stderr, flush=True)
    return float(reduce_max(el))


bench.timed_steps = timed_steps
sys.argv = ["bench.py", "--no-cpu-baseline"]
bench.main()
