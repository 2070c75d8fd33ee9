cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-configs --emulate-world 8 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s  %.3f ms/step' % (d['value'], d['ms_per_step']), '$*')"; }
run
for s in 2 4; do for g in 640 768 1024 1280; do run --opt split_lone_batch=$s --opt trace_grid=$g --opt trace_grid_alone=$g; done; done
