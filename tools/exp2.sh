R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$*')"; }
run
run --opt trace_slices=1
run --opt trace_slices=8
run --opt trace_slices=64
run --opt shade_grid=256
run --opt shade_grid=1024
run --opt shade_grid=2048
for rm in 8 16 24 36; do for nm in 4 8 12 20; do run --opt trace_refill_min=$rm --opt trace_node_min=$nm; done; done
