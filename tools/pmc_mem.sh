# memory-pipeline counters of the traversal kernel (separate rocprofv3 passes, few counters per hardware
# block, each pass under its own timeout): bash tools/pmc_mem.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-mem}
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432"
p() { n=$1; shift; timeout -k 5 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/${T}_$n -- $B > $R/gpurun_out/${T}_$n.log 2>&1; echo "pass $n rc=$?"; }
p a GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
p b TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
p c SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU
p d SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
p e TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum
p f TD_TD_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
p g TCP_GATE_EN1_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
