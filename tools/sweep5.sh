R=$GRAFT_REPO_ROOT
run() { v=$(python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"); echo "$* : $v"; }
for nm in 36 42 48 56; do for rm in 14 18 22; do run --opt trace_node_min=$nm --opt trace_refill_min=$rm; done; done
python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-traffic --opt trace_node_min=40 --opt trace_refill_min=18 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], r['kernel_ms'], r['node_visits_per_ray'], r['prim_tests_per_ray'], r['wave_diag_ordered'])"
