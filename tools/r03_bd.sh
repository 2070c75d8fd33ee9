cd $GRAFT_REPO_ROOT
for o in "" "trace_grid_alone=768" "trace_grid_alone=1024" "trace_grid_alone=1536" "trace_grid_alone=2048" "trace_grid_alone=1280,trace_lds_depth=12" ; do
 for i in 1 2; do echo -n "config5 [$o]: "; timeout 300 python bench.py --configs-only "config5_veach_bdpt_512x512_64spp:$o" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); v=list(d.values())[0]; print(v['seconds'], v['Mrays_per_s'])"; done; done
