rm -f gpurun_out/symbench_*.txt
NDMPS_TRD_XCD=1 bash tools/batch_sweep.sh gpurun_out/symbench_xcd.txt "64 2" "64 2" "64 4" "96 3" "128 4" "64 1" > /dev/null
NDMPS_TRD_XCD=1 NDMPS_TRD_SYM=1 bash tools/batch_sweep.sh gpurun_out/symbench_sym_xcd.txt "64 4" "64 4" "128 4" "96 6" "64 3" > /dev/null
for f in xcd sym_xcd; do echo $f; cat gpurun_out/symbench_$f.txt; done
