R=$GRAFT_REPO_ROOT
run() { v=$(python $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"); echo "$* : $v"; }
for l in 2 3 4 5 6 8; do run --opt overlap_lanes=$l; done
run --opt trace_grid=512; run --opt trace_grid=320; run --opt trace_grid=256 --opt overlap_lanes=6
run --frames-per-step 16 --steps 16; run --frames-per-step 16 --steps 16 --opt overlap_lanes=6
run --opt shade_grid=2048; run --opt shade_grid=512
