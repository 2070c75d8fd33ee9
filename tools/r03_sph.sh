R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r03e_bench.json 2> gpurun_out/r03e_bench.err; tail -1 gpurun_out/r03e_bench.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); r=d['roofline']; print(r['fractions']); print(json.dumps(r['valu'], indent=1))"
