cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_exact
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_exact -- python3 tools/exact_flow_probe.py 256 > gpurun_out/exact_flow.log 2>&1
db=$(ls gpurun_out/prof_exact/*/*_results.db | head -1)
python tools/prof_db.py $db 25 > gpurun_out/exact_flow_kernels.txt 2>&1
rm -rf gpurun_out/prof_exact
grep -v "^W2026\|amdgpu.ids" gpurun_out/exact_flow.log | tail -8
head -30 gpurun_out/exact_flow_kernels.txt | cut -c1-150
