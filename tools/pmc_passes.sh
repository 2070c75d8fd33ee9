# rocprofv3 passes on the GPU box (each counter group in its own run, as gpurun requires):
#   bash tools/pmc_passes.sh <tag>      -> gpurun_out/<tag>_*/...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-pmc}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --opt batch_paths=33554432 --opt merge_paths=33554432"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- $B > $R/gpurun_out/${T}_stats.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_sq -- $B > $R/gpurun_out/${T}_sq.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/${T}_tcc -- $B > $R/gpurun_out/${T}_tcc.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${T}_fetch -- $B > $R/gpurun_out/${T}_fetch.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_write -- $B > $R/gpurun_out/${T}_write.log 2>&1
