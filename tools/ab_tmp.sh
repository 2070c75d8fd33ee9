cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_bdpt.py tests/test_gpu_bdpt_spec.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
for i in 1 2 3; do python tools/bdpt_bench.py 64 512 | cut -c1-100; done
bash tools/pmc_bdpt3.sh 2>&1 | grep -E "resolve"
