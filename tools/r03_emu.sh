cd $GRAFT_REPO_ROOT
run() { O=""; for kv in "$@"; do O="$O --opt $kv"; done
  v=$(timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world 8 $O 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$* : $v"; }
run
for b in 4194304 6291456 8388608 12582912 16777216; do run batch_paths=$b merge_paths=$b; done
run batch_paths=4194304 merge_paths=4194304 overlap_lanes=6
run batch_paths=8388608 merge_paths=8388608 overlap_lanes=3
run batch_paths=2097152 merge_paths=2097152 overlap_lanes=8
