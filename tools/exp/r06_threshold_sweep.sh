R=${GRAFT_REPO_ROOT:-.}
run() { timeout 300 python $R/bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$*')"; }
run; run
for nm in 30 38 46 54; do for rm in 12 18 24 30; do run --opt trace_node_min=$nm --opt trace_refill_min=$rm; done; done
for g in 1024 1280 1536; do run --opt trace_grid_alone=$g; done
for s in 8 16 32 64; do run --opt trace_slices=$s; done
run
