R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
run() { timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$V $*')"; }
b() { make -s -C $R/ti_raytrace_amd/csrc clean; make -s -C $R/ti_raytrace_amd/csrc EXTRA="$1" 2>&1 | grep -E "error"; }
V=""; run; run --opt overlap_lanes=1
V="-DTR_MIN_WAVES=4"; b "$V"; run; run --opt trace_grid=1024; run --opt overlap_lanes=1
V="-DTR_MIN_WAVES=3"; b "$V"; run --opt trace_grid=768; run --opt trace_grid=1536
