cd $GRAFT_REPO_ROOT
run() { O=""; for kv in "$@"; do O="$O --opt $kv"; done
  v=$(timeout 300 python bench.py --no-cpu-baseline --no-roofline $O 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "$* : $v"; }
run
run trace_refill_min=14
run trace_refill_min=22
run trace_refill_min=26
run trace_refill_min=30
run trace_node_min=34
run trace_node_min=42
run trace_refill_min=22 trace_node_min=42
run trace_refill_min=26 trace_node_min=44
run trace_slices=16
run trace_slices=64
run overlap_lanes=3
run overlap_lanes=2
run
