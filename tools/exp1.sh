# scratch experiment driver (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/exp1_trace -- $B --opt overlap_lanes=1 > $R/gpurun_out/exp1_trace.log 2>&1
run() { python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s' % d['value'], '$*')"; }
run
run --opt trace_grid=1024
run --opt trace_grid=1280
run --opt trace_grid=1792
run --opt trace_grid=2048
run --opt trace_grid=1280 --opt overlap_lanes=6
run --opt trace_grid=1024 --opt overlap_lanes=6
