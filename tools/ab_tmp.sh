cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bdpt.py tests/test_gpu_bdpt_spec.py tests/test_gpu_render.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -2
for i in 1 2 3; do python tools/bdpt_bench.py 64 512 | cut -c40-100; done
