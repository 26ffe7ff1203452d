"""Probe (GPU_MAX_HW_QUEUES=8): pairwise interference of streams for (a) long spin kernels (same queue ->
2x) and (b) dependent chains of tiny kernels (dispatch path shared -> slower than alone)."""
import time, torch
torch.cuda.init()
n = 10
streams = [torch.cuda.Stream() for _ in range(n)]
def run(pair, cyc, reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for s in pair:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for label, cyc, reps in (("spin 85us x40", 200000, 40), ("tiny x2000", 2000, 2000)):
    for s in streams: run((s,), cyc, 5)
    base = min(run((streams[0],), cyc, reps) for _ in range(3))
    print(label, "one stream", round(base, 2), "ms")
    for i in range(n):
        row = []
        for j in range(n):
            row.append("  . " if i == j else ("%4.2f" % (run((streams[i], streams[j]), cyc, reps) / base)))
        print(i, " ".join(row), flush=True)
