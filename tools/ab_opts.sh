# A/B builds with bench options: bash tools/ab_opts.sh "<EXTRA flags>|<bench opts>" ...
R=${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  F="${V%%|*}"; O="${V#*|}"
  make -s -C $R/ti_raytrace_amd/csrc clean; make -s -C $R/ti_raytrace_amd/csrc EXTRA="$F" 2>&1 | grep -E "error"
  for i in 1 2; do timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline $O 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$V')"; done
done
