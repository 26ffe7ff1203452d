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


def timed_steps(step, steps, warmup, barrier, reduce_max, after_warmup=None, drain=None):
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

    r = orig(timed_step, steps, warmup, barrier, reduce_max, aw, drain)
    t1 = time.perf_counter()
    inside = [(g, round(ms, 2)) for g, ms, t in events if timed_steps.t0 <= t <= t1]
    print(f"# {steps} steps, {r / steps * 1e3:.2f} ms per step; collections inside the timed region (generation, ms): {inside}",
          file=sys.stderr)
    print("# host time per step (ms, warm-up first): " + " ".join(f"{t:.1f}" for t in times), file=sys.stderr)
    return r


from imgcompressionmps_amd.core import batch as _hb  # noqa: E402
from imgcompressionmps_amd.core import ndmps as _nd  # noqa: E402

halves = []
_begin, _result = _hb.encode_decode_begin, _hb.PendingBatch.result
_gbegin, _gresult = _nd.NDMPS.from_tensors_begin.__func__, _nd.PendingGroup.result


import faulthandler  # noqa: E402


def begin(*a, **k):
    t = time.perf_counter()
    faulthandler.dump_traceback_later(0.4, file=sys.stderr)  # where is the host when an enqueue takes this long?
    try:
        r = _begin(*a, **k)
    finally:
        faulthandler.cancel_dump_traceback_later()
    halves.append(("begin", (time.perf_counter() - t) * 1e3))
    return r


def result(self):
    t = time.perf_counter()
    r = _result(self)
    halves.append(("result", (time.perf_counter() - t) * 1e3))
    return r


def gresult(self):
    t = time.perf_counter()
    r = _gresult(self)
    dt = (time.perf_counter() - t) * 1e3
    if dt > 50:
        halves.append(("group result", dt))
    return r


_hb.encode_decode_begin = begin
_hb.PendingBatch.result = result
_nd.PendingGroup.result = gresult
bench.timed_steps = timed_steps
import atexit  # noqa: E402
atexit.register(lambda: print("# halves > 30 ms: " + " ".join(f"{n}:{ms:.0f}" for n, ms in halves if ms > 30), file=sys.stderr))
sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
