R=${GRAFT_REPO_ROOT:-.}
run() { timeout 300 python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$*')"; }
for nm in 24 32 40 48 56; do for rm in 8 16 24; do run --opt trace_node_min=$nm --opt trace_refill_min=$rm; done; done
