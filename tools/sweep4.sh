R=$GRAFT_REPO_ROOT
run() { v=$(python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"); echo "$* : $v"; }
run; run --opt overlap_lanes=1
for nm in 16 22 28 36 44; do for rm in 8 14 20 28 40; do run --opt trace_node_min=$nm --opt trace_refill_min=$rm --opt overlap_lanes=1; done; done
