# k_trace_q: parity + A/B against k_trace
R=$GRAFT_REPO_ROOT; T=${1:-r03q}
cd $R
timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_gpu_render.py tests/test_gpu_kat.py -q -m gpu -x > gpurun_out/${T}_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/${T}_pytest.log | tail -5
for o in "trace_queue=0" "trace_queue=1" "trace_queue=1 --opt trace_node_min=32" "trace_queue=1 --opt trace_node_min=44" "trace_queue=1 --opt trace_node_min=50" "trace_queue=1 --opt trace_refill_min=10" "trace_queue=1 --opt trace_refill_min=26"; do
  echo -n "$o : "; timeout 300 python bench.py --no-cpu-baseline --no-roofline --opt $o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
done 2>&1 | tee gpurun_out/${T}_ab.log
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel_ms'], r['wave_diag_ordered'], r['node_visits_per_ray'], r['prim_tests_per_ray'])" 2>&1 | tee -a gpurun_out/${T}_ab.log
