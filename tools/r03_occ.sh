cd $GRAFT_REPO_ROOT
run() { L=$1; shift; O=""; for kv in "$@"; do O="$O --opt $kv"; done
  v=$(TIRT_LIB_PATH=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --no-cpu-baseline --no-roofline $O 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "$L $* : $v"; }
G="trace_grid=1280 trace_grid_alone=1280"
A=gpurun_in/libtirt_b256c224.so
run $A $G
run $A $G overlap_lanes=3
run $A $G trace_refill_min=12
run $A $G trace_refill_min=24
run $A $G trace_refill_min=32
run $A $G trace_node_min=32
run $A $G trace_node_min=44
run $A $G trace_node_min=26
run $A $G trace_slices=4
run $A $G trace_slices=6
run gpurun_in/libtirt_b256c240.so $G
run gpurun_in/libtirt_b256c240.so $G trace_lds_depth=15
run gpurun_in/libtirt_b256c288.so $G trace_lds_depth=12
run gpurun_in/libtirt_b256c288.so $G trace_lds_depth=13
run $A $G trace_lds_depth=14
run $A trace_grid=1280 trace_grid_alone=1536
run $A trace_grid=1152 trace_grid_alone=1280
