cd $GRAFT_REPO_ROOT
for n in 1 2 4 8; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world $n 2>/dev/null | tail -1; done > gpurun_out/r03h_emulated_rank_scaling.log
for s in 4 8 16 32; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world 8 --steps $s 2>/dev/null | tail -1; done > gpurun_out/r03h_emulated_8_ranks_steps_4_8_16_32.log
python - <<PY
import json
for f in ("gpurun_out/r03h_emulated_rank_scaling.log","gpurun_out/r03h_emulated_8_ranks_steps_4_8_16_32.log"):
    for l in open(f):
        d=json.loads(l); print(f[-40:], d['steps'], d['value'], d['ms_per_step'])
PY
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_dist.py -q -m gpu 2>&1 | tail -2
