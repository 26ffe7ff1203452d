nv=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_single
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_single -- python3 tools/single_probe.py $nv 256 64 4 > gpurun_out/single_$nv.log 2>&1
db=$(ls gpurun_out/prof_single/*/*_results.db | head -1)
python tools/prof_db.py $db 5 --gaps gram128_kernel 90 > gpurun_out/single_${nv}_timeline.txt 2>&1
rm -rf gpurun_out/prof_single
tail -2 gpurun_out/single_$nv.log
