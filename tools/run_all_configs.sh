#!/bin/bash
# Every bench line of the round: tools/run_all_configs.sh TAG  -> gpurun_out/TAG_bench_{metric,2,3,4,5}.json
tag=$1
for c in metric 2 3 4 5; do
  python bench.py --config $c > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err || echo "config $c FAILED: $(tail -3 gpurun_out/${tag}_bench_$c.err)"
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_bench_$c.json"))
    r = d["roofline"]
    print("config $c: %.0f Mvox/s  %.2f ms/step  roofline %s %.3f (%s)" % (d["value"], d["ms_per_step"], r["bound"], r["frac"] or 0, (r["kernel"] or "")[:40]),
          "parity", {k: (round(v, 9) if isinstance(v, float) else v) for k, v in d.get("parity", {}).items() if k in ("ssim_gap", "rel_frobenius_vs_oracle", "bonds_equal")})
except Exception as e:
    print("config $c: no line (%s)" % e)
PY
done
