R=$GRAFT_REPO_ROOT
run() { v=$(python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"); echo "$* : $v"; }
for depth in 16 20 24; do run --opt trace_lds_depth=$depth; run --opt trace_lds_depth=$depth --opt overlap_lanes=1; done
run --opt trace_lds_depth=32 --opt trace_grid_alone=256 --opt trace_grid=256
run --opt trace_lds_depth=32 --opt trace_grid_alone=256 --opt trace_grid=256 --opt overlap_lanes=1
