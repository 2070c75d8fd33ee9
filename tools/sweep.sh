# usage: bash tools/sweep.sh  -- quick tuning sweep of the traversal knobs (GPU box)
R=${GRAFT_REPO_ROOT:-.}
run() { python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$*')"; }
run
for nm in 4 8 12 16 24; do for rm in 24 32 40; do run --opt trace_node_min=$nm --opt trace_refill_min=$rm; done; done
run --opt trace_node_min=12 --opt trace_refill_min=32 --opt trace_grid=1024
run --opt trace_node_min=12 --opt trace_refill_min=32 --opt trace_grid=1280
run --opt trace_node_min=12 --opt trace_refill_min=32 --opt trace_lds_depth=32
run --opt trace_node_min=12 --opt trace_refill_min=32 --opt trace_lds_depth=20
