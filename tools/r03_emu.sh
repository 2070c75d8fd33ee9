cd $GRAFT_REPO_ROOT
run() { W=$1; shift; S=$1; shift; O=""; for kv in "$@"; do O="$O --opt $kv"; done
  v=$(timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world $W --steps $S $O 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "world $W steps $S $* stagger=$TIRT_STAGGER : $v"; }
run 8 8
run 8 8 split_lone_batch=2
export TIRT_STAGGER=0
run 8 8 split_lone_batch=2
export TIRT_STAGGER=1
run 8 8 split_lone_batch=2
export TIRT_STAGGER=3
run 8 8 split_lone_batch=2
export TIRT_STAGGER=0
run 8 8 split_lone_batch=3
run 8 8 split_lone_batch=4
