# A/B of libtirt.so build variants, each with its own context options, in ONE gpurun call:
#   bash tools/ab5.sh <tag> "name|EXTRA flags|bench args" ...      (env PRE="<shell command>" runs first on the GPU box)
# Variants with the same flags share a library (built once, HERE, cross-compiled); results -> gpurun_out/<tag>_ab.log
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
mkdir -p $R/ab_libs $R/gpurun_out
declare -A LIBOF
CMD="${PRE:-true}; "
n=0
for spec in "$@"; do
  IFS='|' read -r name flags opts <<< "$spec"
  key=$(echo "$flags" | md5sum | cut -c1-10)
  if [ -z "${LIBOF[$key]}" ]; then
    LIBOF[$key]=1
    ( make -s -j2 -C $R/ti_raytrace_amd/csrc OUT=$R/ab_libs/v$key.so OBJ=$R/ab_obj/v$key EXTRA="$flags" 2>&1 | grep -E "error" || true ) &
    n=$((n+1)); if [ $((n % 4)) -eq 0 ]; then wait; fi
  fi
  ONE="python -c \"import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-14s %8.1f Mrays/s  %8.4f ms/step  rays %d' % (sys.argv[1], d['value'], d['ms_per_step'], d['rays']['closest'] + d['rays']['shadow']))\""
  CMD="$CMD for i in 1 2; do TIRT_LIB_PATH=\$GRAFT_REPO_ROOT/ab_libs/v$key.so timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 8 --warmup 1 $opts 2>&1 | tail -1 | $ONE $name || echo $name FAILED; done; "
done
wait
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-2400} -- "$CMD" > $R/gpurun_out/${TAG}_ab.log 2>&1
grep -E "Mrays/s|FAILED|status=" $R/gpurun_out/${TAG}_ab.log
