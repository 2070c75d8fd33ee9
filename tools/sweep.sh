# usage: bash tools/sweep.sh  -- quick tuning sweep of the traversal knobs (GPU box); results of round 2: profiles/r02*_sweep*.log
R=${GRAFT_REPO_ROOT:-.}
run() { timeout 300 python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$*')"; }
run
for nm in 30 38 46; do for rm in 12 18 24; do run --opt trace_node_min=$nm --opt trace_refill_min=$rm; done; done
for d in 12 14 16 20; do run --opt trace_lds_depth=$d; done
for g in 320 384 512; do run --opt trace_grid=$g; done
for l in 1 2 3 4 6; do run --opt overlap_lanes=$l; done
for w in 2 4 8; do run --emulate-world $w; done
