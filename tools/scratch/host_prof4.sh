#!/bin/bash
python - <<'PY' 2>&1 | tail -45
import cProfile, pstats, sys, io, threading
sys.argv = ["bench.py", "--config", "4", "--steps", "40", "--skip-single", "--no-cpu-baseline", "--no-configs"]
import bench
prof = cProfile.Profile()
# profile the worker threads too
threading.setprofile(lambda *a: None)
prof.enable()
bench.main()
prof.disable()
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(25)
print(s.getvalue()[:5000])
PY
