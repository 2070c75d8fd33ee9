# bash tools/sweep2.sh  -- option sweep of the traversal launch geometry (GPU box)
R=$GRAFT_REPO_ROOT
for depth in 12 14 16; do for alone in 512 768; do for grid in 384 512 768; do
  v=$(python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --opt trace_lds_depth=$depth --opt trace_grid_alone=$alone --opt trace_grid=$grid 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "depth $depth alone $alone grid $grid : $v"
done; done; done
for lanes in 1 2 3 4 6; do
  v=$(python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --opt overlap_lanes=$lanes --opt trace_lds_depth=14 --opt trace_grid_alone=768 --opt trace_grid=512 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "lanes $lanes : $v"
done
