# GPU pass: parity suite + default bench + the N>1 flow on one rank (RCCL) and on three ranks sharing the GPU (gloo)
R=$GRAFT_REPO_ROOT; T=${1:-r03d}
cd $R
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/${T}_pytest_gpu.log | tail -3
( time timeout 900 python bench.py ) > gpurun_out/${T}_bench_default.log 2>&1; tail -c 1200 gpurun_out/${T}_bench_default.log
TIRT_FORCE_DIST=1 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${T}_bench_force_dist.log 2>&1; tail -1 gpurun_out/${T}_bench_force_dist.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force_dist', d['value'], d.get('distributed'))"
