cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/tools/run_configs.py 5"
p() { n=$1; shift; timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/bd_$n -- $B > $R/gpurun_out/bd_$n.log 2>&1; echo "pass $n rc=$?"; }
p a GRBM_GUI_ACTIVE TA_TA_BUSY_sum
p c SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
p d SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
