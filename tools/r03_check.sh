# GPU pass: parity suite + default bench (with configs + PMC passes)
R=$GRAFT_REPO_ROOT; T=${1:-r03c}
cd $R
timeout 1800 python -m pytest tests -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/${T}_pytest_gpu.log | tail -3
( time timeout 900 python bench.py ) > gpurun_out/${T}_bench_default.log 2>&1; tail -c 3000 gpurun_out/${T}_bench_default.log
