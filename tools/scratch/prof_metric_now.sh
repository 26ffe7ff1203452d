#!/bin/bash
# kernel trace of the headline's timed loop: per-kernel table, phase times and idle gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-now}
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python3 bench.py --config metric --skip-single --no-cpu-baseline --no-configs > gpurun_out/prof_$tag.json 2>/dev/null
db=$(ls gpurun_out/prof_$tag/*/*_results.db | head -1)
python tools/prof_db.py $db 30 --region gpurun_out/prof_$tag.json > gpurun_out/${tag}_kernels.txt 2>&1
python tools/phase_time.py $db > gpurun_out/${tag}_phase.txt 2>&1
python tools/gap_report.py $db 12 --region gpurun_out/prof_$tag.json > gpurun_out/${tag}_gaps.txt 2>&1
rm -rf gpurun_out/prof_$tag
cut -c1-150 gpurun_out/${tag}_kernels.txt | head -24; head -12 gpurun_out/${tag}_phase.txt; head -8 gpurun_out/${tag}_gaps.txt
