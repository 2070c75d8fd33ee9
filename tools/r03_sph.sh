R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03sph_pytest.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r03sph_pytest.log | tail -8
timeout 300 python tools/timeline.py 1 2>&1 | grep -E "launch|alive"
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err; tail -1 gpurun_out/r03e_bench.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['kernel_ms'], r['fractions'], r['valu']); print({k:(v.get('value'), v.get('seconds')) for k,v in d['configs'].items() if isinstance(v,dict)})"
