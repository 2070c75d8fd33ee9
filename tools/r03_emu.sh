cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_spectral.py tests/test_gpu_bench_ranks.py tests/test_gpu_gallery.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
run() { W=$1; shift; S=$1; shift; O=""; for kv in "$@"; do O="$O --opt $kv"; done
  v=$(timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world $W --steps $S $O 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "world $W steps $S $* : $v"; }
run 1 1
run 1 1 pair_batches=0
run 1 8
run 1 8 pair_batches=0
run 8 8
run 8 8 pair_batches=0
run 4 8
run 4 8 pair_batches=0
run 2 8
run 2 8 pair_batches=0
