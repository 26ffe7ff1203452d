tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_${tag}
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag} -- python3 bench.py --skip-single --no-cpu-baseline > /dev/null 2>&1
db=$(ls gpurun_out/prof_${tag}/*/*_results.db | head -1)
python tools/prof_db.py $db 40 > gpurun_out/${tag}_rocprof_kernel_stats_config_metric_timed_loop.txt 2>&1
python tools/phase_time.py $db > gpurun_out/${tag}_phase_time_bench_timed_loop.txt 2>&1
python tools/overlap.py $db > gpurun_out/${tag}_overlap.txt 2>&1
rm -rf gpurun_out/prof_${tag}
cat gpurun_out/${tag}_rocprof_kernel_stats_config_metric_timed_loop.txt | head -50
cat gpurun_out/${tag}_phase_time_bench_timed_loop.txt | head -40
