"""Scratch: do Python garbage collections land inside bench.py's timed region, and what do they cost?  Logs every
collection (generation, duration) between the end of the warm-up and the closing barrier; GCFREEZE=1 collects and freezes
the heap after the warm-up (gc.freeze: later full collections only look at what was allocated since).
usage: python tools/scratch/gc_probe.py [bench.py arguments]"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

events = []


def cb(phase, info):
    if phase == "start":
        cb.t = time.perf_counter()
    else:
        events.append((info["generation"], (time.perf_counter() - cb.t) * 1e3, time.perf_counter()))


gc.callbacks.append(cb)
orig = bench.timed_steps


def timed_steps(step, steps, warmup, barrier, reduce_max, after_warmup=None):
    def aw():
        if after_warmup is not None:
            after_warmup()
        if os.environ.get("GCFREEZE"):
            gc.collect()
            gc.freeze()
        timed_steps.t0 = time.perf_counter()

    times = []

    def timed_step():
        t = time.perf_counter()
        step()
        times.append((time.perf_counter() - t) * 1e3)

    r = orig(timed_step, steps, warmup, barrier, reduce_max, aw)
    t1 = time.perf_counter()
    inside = [(g, round(ms, 2)) for g, ms, t in events if timed_steps.t0 <= t <= t1]
    print(f"# {steps} steps, {r / steps * 1e3:.2f} ms per step; collections inside the timed region (generation, ms): {inside}",
          file=sys.stderr)
    print("# host time per step (ms, warm-up first): " + " ".join(f"{t:.1f}" for t in times), file=sys.stderr)
    return r


bench.timed_steps = timed_steps
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
