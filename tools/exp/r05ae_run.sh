cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bdpt.py tests/test_gpu_bdpt_spec.py tests/test_refkat.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
for i in 1 2 3; do python tools/bdpt_bench.py 64 512; done
python tools/bdpt_bench.py 64 512 overlap_lanes=1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05ae_bd -- python $GRAFT_REPO_ROOT/tools/bdpt_bench.py 64 512 overlap_lanes=1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r05ae_bd -name "*kernel_stats.csv" | head -1); cut -c1-60,200- $f | head -12; cut -d, -f1-4 $f | head -12 | cut -c1-40,100-
