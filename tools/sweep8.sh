R=$GRAFT_REPO_ROOT
run() { v=$(python $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"); echo "$* : $v"; }
for w in 0 2 4 8; do run --emulate-world $w --opt split_lone_batch=0; run --emulate-world $w --opt split_lone_batch=1; done
