# full measurement pass on the GPU box: bash tools/measure_all.sh <tag>   (outputs under gpurun_out/<tag>_*)
R=$GRAFT_REPO_ROOT; T=${1:-rXX}
cd $R
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -1 gpurun_out/${T}_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/${T}_bench_default.log 2>&1; tail -c 600 gpurun_out/${T}_bench_default.log
for n in 1 2 4 8; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world $n 2>&1 | tail -1; done > gpurun_out/${T}_emulated_rank_scaling.log
timeout 600 python tools/run_configs.py > gpurun_out/${T}_configs.log 2>&1
timeout 900 python tools/big_scene_check.py 1000000 > gpurun_out/${T}_big1m.log 2>&1
timeout 900 python tools/big_scene_check.py 4000000 > gpurun_out/${T}_big4m.log 2>&1
bash tools/pmc_passes.sh $T
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats1lane -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432 > $R/gpurun_out/${T}_stats1lane.log 2>&1
cd $R; bash tools/pmc_mem.sh ${T}mem
echo done
