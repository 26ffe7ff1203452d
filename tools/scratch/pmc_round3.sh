cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_f2 gpurun_out/pmc_w2 gpurun_out/pmc_f3 gpurun_out/pmc_w3
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 tools/gram_probe.py 32768 512 3 32 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 tools/gram_probe.py 32768 512 3 32 > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc_f gpurun_out/pmc_w "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 tools/gram_probe.py 32768 512 3 32  (32 matrices of 32768 x 512 fp32 per launch: one lockstep group's raw Gram; build of r03_k)" gpurun_out/r03_pmc_summary.json gram128_kernel,gram128_reduce_kernel
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f2 -- python3 tools/gram_gather_probe.py 32 3 64 sorted > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w2 -- python3 tools/gram_gather_probe.py 32 3 64 sorted > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc_f2 gpurun_out/pmc_w2 "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 tools/gram_gather_probe.py 32 3 64 sorted  (the 64-column raw Gram of a lockstep group of 32 read through the tables, rows in memory order: what the fused sweep launches)" gpurun_out/r03_pmc_summary.json gram64_stream_kernel,tile_reduce_batched_kernel
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f3 -- python3 tools/scratch/proj64_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w3 -- python3 tools/scratch/proj64_probe.py > /dev/null 2>&1
python tools/pmc_to_json.py gpurun_out/pmc_f3 gpurun_out/pmc_w3 "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 tools/scratch/proj64_probe.py  (first projection of a bond cap of 32, 32 volumes)" gpurun_out/r03_pmc_summary.json proj64_stream_kernel
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/pmc_f2 gpurun_out/pmc_w2 gpurun_out/pmc_f3 gpurun_out/pmc_w3
cat gpurun_out/r03_pmc_summary.json | head -80
