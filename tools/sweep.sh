# usage: bash tools/sweep.sh  -- quick tuning sweep of the traversal knobs (GPU box)
R=${GRAFT_REPO_ROOT:-.}
run() { timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$*')"; }
run
for nm in 20 28 36; do for rm in 12 20 28; do run --opt trace_node_min=$nm --opt trace_refill_min=$rm; done; done
run --opt trace_lds_depth=14
run --opt trace_lds_depth=18
run --opt trace_grid=768
run --opt trace_grid=1024
run --opt overlap_lanes=3
run --opt overlap_lanes=6
run --opt shade_grid=1024
