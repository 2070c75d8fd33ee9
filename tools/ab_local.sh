# Build A/B variants of libtirt.so HERE (cross-compile, in parallel) and time them on the GPU box in one gpurun call:
#   bash tools/ab_local.sh <tag> "<bench args>" name1="<EXTRA flags>" name2="<EXTRA flags>" ...
# Variant libraries go to ab_libs/<name>.so (git-ignored, travels with gpurun); bench.py loads them through TIRT_LIB_PATH.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift; BARGS=$1; shift
mkdir -p $R/ab_libs $R/gpurun_out
NAMES=""
for kv in "$@"; do
  name=${kv%%=*}; flags=${kv#*=}
  ( make -s -j4 -C $R/ti_raytrace_amd/csrc OUT=$R/ab_libs/$name.so OBJ=$R/ab_obj/$name EXTRA="$flags" 2>&1 | grep -E "error" || true ) &
  NAMES="$NAMES $name"
done
wait
CMD="for n in $NAMES; do for i in 1 2; do TIRT_LIB_PATH=\$GRAFT_REPO_ROOT/ab_libs/\$n.so timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-configs $BARGS 2>&1 | tail -1 | python -c \"import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-24s %8.1f Mrays/s  %.4f ms/step' % ('\$n', d['value'], d['ms_per_step']))\"; done; done"
/usr/local/graft/bin/gpurun --timeout 1500 -- "$CMD" > $R/gpurun_out/${TAG}_ab.log 2>&1
grep -E "Mrays/s|status=" $R/gpurun_out/${TAG}_ab.log
