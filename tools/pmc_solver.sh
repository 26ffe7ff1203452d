#!/bin/bash
# PMC pass over the latency-bound solver kernels: tools/pmc_solver.sh OUT.json
out=${1:-gpurun_out/r04_pmc_summary.json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
rm -rf gpurun_out/pmc_s1 gpurun_out/pmc_s2 gpurun_out/pmc_s3
rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_s1 -- python3 tools/trd_probe.py 32 512 64 > gpurun_out/pmc_s1.log 2>&1; echo pass 1 done
python tools/pmc_solver_to_json.py gpurun_out/pmc_s1 "rocprofv3 --pmc $C --output-format csv -- python3 tools/trd_probe.py 32 512 64  (32 order-512 Gram matrices, 64 vectors each: the eigenproblems of a lockstep group of the metric configuration)" $out trd_team_kernel,trd_tail_kernel,trd_bisect_kernel,trd_invit_kernel,trd_ortho_kernel,trd_back_kernel,blk_step_kernel > /dev/null
rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_s2 -- python3 tools/solver_once.py 2048 128 > gpurun_out/pmc_s2.log 2>&1; echo pass 2 done
python tools/pmc_solver_to_json.py gpurun_out/pmc_s2 "rocprofv3 --pmc $C --output-format csv -- python3 tools/solver_once.py 2048 128  (one order-2048 Gram matrix, 128 vectors: BASELINE config 5's eigenproblem; the matrix whole in the resident kernel, eigenvectors across the chip)" $out trd_team_kernel,trd_tail_reg_kernel,trd_bisect_kernel,trd_invit_kernel,back_rows_step_kernel,back_wide_t_kernel,wide_chol_diag_kernel " @ order 2048" > /dev/null
rm -rf gpurun_out/pmc_s4
rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_s4 -- python3 tools/solver_once.py 4096 128 > gpurun_out/pmc_s4.log 2>&1; echo pass 2b done
python tools/pmc_solver_to_json.py gpurun_out/pmc_s4 "rocprofv3 --pmc $C --output-format csv -- python3 tools/solver_once.py 4096 128  (one order-4096 Gram matrix, 128 vectors: 2048 columns on the panel-blocked launches, the last 2048 in the resident kernel)" $out pnl_vec_kernel,pnl_symv_kernel,pnl_update_kernel > /dev/null
rm -rf gpurun_out/pmc_s4
rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_s3 -- python3 tools/solver_once.py 4096 4096 > gpurun_out/pmc_s3.log 2>&1; echo pass 3 done
python tools/pmc_solver_to_json.py gpurun_out/pmc_s3 "rocprofv3 --pmc $C --output-format csv -- python3 tools/solver_once.py 4096 4096  (every eigenpair of an order-4096 Gram matrix: the largest site of an exact 256^3 sweep)" $out wide_trsm_kernel,wide_chol_diag_kernel,wide_chol_trail_kernel,back_wide_kernel
rm -rf gpurun_out/pmc_s1 gpurun_out/pmc_s2 gpurun_out/pmc_s3
