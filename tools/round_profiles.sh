#!/bin/bash
# Everything profiles/ holds for a round, on the GPU box: tools/round_profiles.sh TAG
#   gpurun_out/TAG_gpu_tests.log, TAG_bench_{metric,2,3,4,5}.json, TAG_rocprof_*_.txt, TAG_phase_time.txt
# The per-config summaries keep the dispatches that start inside bench.py's timed region (its line carries the region on
# the host clocks, tools/prof_db.py --region): no volume generator, no warm-up, no reference points.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/${tag}_gpu_tests.log 2>&1; tail -3 gpurun_out/${tag}_gpu_tests.log
tools/run_all_configs.sh $tag
for c in metric 2 3 4 5; do
  rm -rf gpurun_out/prof_${tag}_$c gpurun_out/prof_${tag}_$c.json
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_$c -- python3 bench.py --config $c --skip-single --no-cpu-baseline > gpurun_out/prof_${tag}_$c.json 2>/dev/null
  db=$(ls gpurun_out/prof_${tag}_$c/*/*_results.db | head -1)
  python tools/prof_db.py $db 30 --region gpurun_out/prof_${tag}_$c.json > gpurun_out/${tag}_rocprof_kernel_stats_config_${c}_timed_loop.txt 2>&1
  if [ $c = metric ]; then python tools/phase_time.py $db > gpurun_out/${tag}_phase_time_bench_timed_loop.txt 2>&1; fi
  rm -rf gpurun_out/prof_${tag}_$c gpurun_out/prof_${tag}_$c.json
  echo "profiled config $c"
done
rm -rf gpurun_out/prof_${tag}_full
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_full -- python3 bench.py --no-cpu-baseline > /dev/null 2>&1
db=$(ls gpurun_out/prof_${tag}_full/*/*_results.db | head -1)
python tools/prof_db.py $db 30 > gpurun_out/${tag}_rocprof_kernel_stats_bench_default.txt 2>&1
rm -rf gpurun_out/prof_${tag}_full
head -12 gpurun_out/${tag}_rocprof_kernel_stats_config_metric_timed_loop.txt
cat gpurun_out/${tag}_phase_time_bench_timed_loop.txt | head -30
