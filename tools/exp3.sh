R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
run() { timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$*')"; }
run
run --opt overlap_lanes=1
run --opt trace_refill_min=24
run --opt trace_refill_min=16 --opt trace_node_min=8
run --opt trace_refill_min=24 --opt trace_node_min=8
run --opt trace_refill_min=36 --opt trace_node_min=8
run --opt trace_refill_min=36 --opt trace_node_min=16
run --opt trace_refill_min=44 --opt trace_node_min=12
run --opt trace_grid=2048
run --opt trace_grid=1024
