# round-3 first GPU pass: parity suite, bench, refill_min sweep, far-origin stress subset
R=$GRAFT_REPO_ROOT; T=${1:-r03a}
cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -3 gpurun_out/${T}_pytest_gpu.log
timeout 600 python bench.py --no-traffic > gpurun_out/${T}_bench_default.log 2>&1; tail -c 1500 gpurun_out/${T}_bench_default.log
for rm in 4 8 12 16 20; do for nm in 32 38 44; do
  echo -n "refill_min=$rm node_min=$nm : "; timeout 300 python bench.py --no-cpu-baseline --no-roofline --opt trace_refill_min=$rm --opt trace_node_min=$nm 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
done; done > gpurun_out/${T}_sweep_refill.log 2>&1
cat gpurun_out/${T}_sweep_refill.log
timeout 900 python tools/stress_ordered_vs_exhaustive.py 200000 > gpurun_out/${T}_stress.log 2>&1; tail -20 gpurun_out/${T}_stress.log
