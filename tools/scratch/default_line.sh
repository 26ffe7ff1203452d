#!/bin/bash
t0=$(date +%s.%N)
python bench.py > gpurun_out/default_line.json 2> gpurun_out/default_line.err
t1=$(date +%s.%N)
echo "wall $(echo "$t1 - $t0" | bc) s"
tail -2 gpurun_out/default_line.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/default_line.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["unit"], d["n_gpus"], d["steps"], d["warmup"], d["ms_per_step"], d["scaling"], d["dtype"], d["vs_baseline"])
print(d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print({k:(round(v["ms_per_step"],2), round(v["Mvoxels_per_s"])) for k,v in d["configs"].items()})
print(d.get("parity"))
PY
