# A/B builds of libtirt.so on the GPU box: bash tools/ab4.sh "<EXTRA flags>" "<EXTRA flags>" ... (two bench runs each)
R=${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  make -s -C $R/ti_raytrace_amd/csrc clean; make -s -C $R/ti_raytrace_amd/csrc EXTRA="$V" 2>&1 | grep -E "error"
  for i in 1 2; do timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], 'EXTRA=$V')"; done
done
