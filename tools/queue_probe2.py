"""Probe: pairwise serialisation of torch pool streams (spin kernels interleaved on two streams)."""
import time, torch
torch.cuda.init()
n = 12
streams = [torch.cuda.Stream() for _ in range(n)]
cyc = 200000
def run(pair, reps=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for s in pair:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cyc)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
run((streams[0],)); base = run((streams[0],))
print("one stream", round(base, 2), "ms")
for rnd in range(2):
    for i in range(n):
        row = []
        for j in range(n):
            row.append(" . " if i == j else ("%3.1f" % (run((streams[i], streams[j])) / base)))
        print(i, " ".join(row), flush=True)
