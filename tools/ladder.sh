# Bound ladder for k_trace: what limits the traversal kernel is asked of the kernel itself.
#   bash tools/ladder.sh <tag>
# Builds variants of libtirt.so HERE (cross-compile) -- TR_PAD=k identity VALU instructions per node visit (behind the loads / behind
# the sort), TR_PADG=k dummy node-record gathers per visit (tirt_render.hip, TR_LADDER_PADS) -- and times each on the GPU box in ONE
# gpurun call: the headline job (4 overlapped batches) and the same job on one render lane (launches back to back: ms/step = the
# sum of the kernels' durations).  Output: gpurun_out/<tag>_ladder.log; the table goes to profiles/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r05_ladder}
mkdir -p $R/ab_libs $R/gpurun_out
VARIANTS=(
  "base="
  "pad8=-DTR_PAD=8" "pad16=-DTR_PAD=16" "pad32=-DTR_PAD=32" "pad64=-DTR_PAD=64"
  "padl16=-DTR_PAD=16 -DTR_PAD_LATE" "padl32=-DTR_PAD=32 -DTR_PAD_LATE"
  "padg1=-DTR_PADG=1" "padg2=-DTR_PADG=2" "padg4=-DTR_PADG=4" "padg8=-DTR_PADG=8"
  "asmfetch=-DTR_ASM_FETCH" "noearly=-DTR_NO_EARLY_LDS_ADDR"
)
NAMES=""
run=0
for kv in "${VARIANTS[@]}"; do
  name=${kv%%=*}; flags=${kv#*=}
  ( make -s -j2 -C $R/ti_raytrace_amd/csrc OUT=$R/ab_libs/$name.so OBJ=$R/ab_obj/$name EXTRA="$flags" 2>&1 | grep -E "error" || true ) &
  NAMES="$NAMES $name"
  run=$((run+1)); if [ $((run % 4)) -eq 0 ]; then wait; fi
done
wait
ONE="python -c \"import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-10s %-6s %8.1f Mrays/s  %8.4f ms/step  rays %d' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d['rays']['closest'] + d['rays']['shadow']))\""
PRE="(timeout 900 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_trace.py -x -q 2>&1 | tail -5; cd /tmp; \$GRAFT_REPO_ROOT/tools/micro/valu_issue) > gpurun_out/${TAG}_pre.log 2>&1; "
CMD="$PRE for n in $NAMES; do for i in 1 2; do TIRT_LIB_PATH=\$GRAFT_REPO_ROOT/ab_libs/\$n.so timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 8 --warmup 1 2>&1 | tail -1 | $ONE \$n lanes4; done; TIRT_LIB_PATH=\$GRAFT_REPO_ROOT/ab_libs/\$n.so timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 4 --warmup 1 --opt overlap_lanes=1 2>&1 | tail -1 | $ONE \$n lane1; done"
/usr/local/graft/bin/gpurun --timeout 2400 -- "$CMD" > $R/gpurun_out/${TAG}_ladder.log 2>&1
grep -E "Mrays/s|status=|exit" $R/gpurun_out/${TAG}_ladder.log
