R=${GRAFT_REPO_ROOT:-.}
b() { make -s -C $R/ti_raytrace_amd/csrc clean; make -s -C $R/ti_raytrace_amd/csrc EXTRA="$1" 2>&1 | grep -E "error"; }
run() { timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$V $*')"; }
V="-DTR_SCALAR_SLAB -DTR_MIN_WAVES=5"; b "$V"
run --opt trace_grid=1280
run --opt trace_grid=1536
run --opt trace_refill_min=24
run --opt trace_refill_min=44
run --opt trace_node_min=8
run --opt trace_node_min=16
run --opt shade_grid=1024
V="-DTR_SCALAR_SLAB -DTR_MIN_WAVES=7"; b "$V"
run --opt trace_lds_depth=20 --opt trace_grid=1792
run --opt trace_lds_depth=16 --opt trace_grid=1792
V="-DTR_SCALAR_SLAB -DTR_MIN_WAVES=8"; b "$V"
run --opt trace_lds_depth=16 --opt trace_grid=2048
V="-DTR_SCALAR_SLAB -DTR_MIN_WAVES=4"; b "$V"
run --opt trace_grid=1024
run --opt trace_grid=1536
