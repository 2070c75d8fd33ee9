R=$GRAFT_REPO_ROOT; cd $R
for o in "bdpt_stagger=0" "" "trace_grid_alone=1024" "trace_grid_alone=768" "bdpt_batch_items=8388608" "bdpt_batch_items=8388608 trace_grid_alone=1024" "bdpt_batch_items=4194304" "bdpt_batch_items=33554432"; do for i in 1 2; do echo -n "[$o] "; python tools/bdpt_bench.py 64 512 $o | cut -c1-90; done; done
