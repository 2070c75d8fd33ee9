# on the GPU box: bash tools/ab_opts2.sh "<bench args>" "opt=val opt=val" "opt=val" ...   (one bench run per option set, two repeats)
R=${GRAFT_REPO_ROOT:-.}; BARGS=$1; shift
for O in "$@"; do
  OPTS=""; for kv in $O; do OPTS="$OPTS --opt $kv"; done
  for i in 1 2; do timeout 60 python $R/bench.py --no-cpu-baseline --no-roofline --no-configs $BARGS $OPTS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s %8.1f Mrays/s  %.4f ms/step' % ('$O', d['value'], d['ms_per_step']))"; done
done
