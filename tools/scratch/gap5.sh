#!/bin/bash
# where is the GPU idle in a config-5 step? kernel trace of the timed loop, gaps by (kernel before -> kernel after)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
c=${1:-5}
rm -rf gpurun_out/prof_gap_$c
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gap_$c -- python3 bench.py --config $c --skip-single --no-cpu-baseline --no-configs > gpurun_out/prof_gap_$c.json 2>/dev/null
db=$(ls gpurun_out/prof_gap_$c/*/*_results.db | head -1)
python tools/gap_report.py $db 25 --region gpurun_out/prof_gap_$c.json > gpurun_out/gap_report_config_$c.txt 2>&1
python tools/prof_db.py $db 40 --region gpurun_out/prof_gap_$c.json > gpurun_out/gap_kernels_config_$c.txt 2>&1
rm -rf gpurun_out/prof_gap_$c
cat gpurun_out/gap_report_config_$c.txt
