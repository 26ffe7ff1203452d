timeout -k 10 600 python -m pytest tests/test_gpu_robustness.py -x -q -m gpu 2>&1 | tail -5 || exit 1
bash tools/scratch/symprobe.sh
rm -f gpurun_out/symbench_*.txt
NDMPS_TRD_SYM=1 bash tools/batch_sweep.sh gpurun_out/symbench_sym.txt "64 2" "64 1" > /dev/null
bash tools/batch_sweep.sh gpurun_out/symbench_default.txt "64 2" "32 1" "8 1" > /dev/null
echo default; cat gpurun_out/symbench_default.txt; echo sym; cat gpurun_out/symbench_sym.txt
