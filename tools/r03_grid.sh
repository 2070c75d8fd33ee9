R=$GRAFT_REPO_ROOT; cd $R
for o in "trace_grid_alone=768" "trace_grid_alone=704" "trace_grid_alone=640" "trace_grid_alone=576" "trace_grid_alone=512" "trace_grid_alone=448" "trace_grid_alone=512 --opt shade_grid=2048" "trace_grid_alone=640 --opt shade_grid=2048"; do
  for i in 1 2; do echo -n "$o : "; timeout 300 python bench.py --no-cpu-baseline --no-roofline --opt $o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"; done
done 2>&1 | tee gpurun_out/r03_grid_ab.log
