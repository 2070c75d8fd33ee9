# A/B of libtirt.so build variants on BASELINE config 5 (BDPT), in ONE gpurun call: per variant the job's Mrays/s (two lanes, 3 runs) and the one-lane kernel times
#   bash tools/ab_bdpt.sh <tag> "name|EXTRA flags" ...        results -> gpurun_out/<tag>_ab.log
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
mkdir -p $R/ab_libs $R/gpurun_out
CMD="cd /tmp; export TMPDIR=/tmp; "
n=0
for spec in "$@"; do
  IFS='|' read -r name flags <<< "$spec"
  key=$(echo "$flags" | md5sum | cut -c1-10)
  ( make -s -j2 -C $R/ti_raytrace_amd/csrc OUT=$R/ab_libs/v$key.so OBJ=$R/ab_obj/v$key EXTRA="$flags" 2>&1 | grep -E "error" || true ) &
  n=$((n+1)); if [ $((n % 4)) -eq 0 ]; then wait; fi
  L="TIRT_LIB_PATH=\$GRAFT_REPO_ROOT/ab_libs/v$key.so"
  CMD="$CMD echo == $name; for i in 1 2 3; do $L timeout 200 python \$GRAFT_REPO_ROOT/tools/bdpt_bench.py 64 512 | cut -c40-100; done; rm -rf /tmp/abp; $L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -- python \$GRAFT_REPO_ROOT/tools/bdpt_bench.py 64 512 overlap_lanes=1 > /dev/null 2>&1; python \$GRAFT_REPO_ROOT/tools/kstat_short.py /tmp/abp; "
done
wait
/usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-2400} -- "$CMD" > $R/gpurun_out/${TAG}_ab.log 2>&1
grep -E "^==|Mrays|k_bd|k_trace|status=" $R/gpurun_out/${TAG}_ab.log
