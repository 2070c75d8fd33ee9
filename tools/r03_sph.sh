R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_trace.py tests/test_gpu_render.py tests/test_gpu_bdpt.py tests/test_gpu_spectral.py tests/test_gpu_gallery.py -q -m gpu -x > gpurun_out/r03sph_pytest.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r03sph_pytest.log | tail -8
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-configs --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['kernel_ms'])"; done
