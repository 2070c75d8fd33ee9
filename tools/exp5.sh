R=${GRAFT_REPO_ROOT:-.}
run() { timeout 300 python $R/bench.py --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%8.1f Mrays/s %7.3f ms/step' % (d['value'], d['ms_per_step']), '$*')"; }
for n in 8 4 2; do
run --emulate-world $n
run --emulate-world $n --opt merge_paths=16777216
run --emulate-world $n --opt merge_paths=33554432
run --emulate-world $n --opt merge_paths=33554432 --opt batch_paths=16777216
run --emulate-world $n --opt merge_paths=33554432 --opt batch_paths=8388608
done
