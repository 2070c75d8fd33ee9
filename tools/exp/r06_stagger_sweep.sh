# 8 emulated ranks at BASELINE's 256 spp: batch plans and staggered starts.  bash tools/r06_stagger.sh
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world ${W:-8} "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms/step  %.1f Mrays/s' % (d['ms_per_step'], d['value']))"; }
for rep in 1 2; do
echo "default (2 x 16 Mi, lockstep)"; run
for nb in 2 4 8; do for sb in 0 1 2 3 4 6; do for ln in 2 4; do
  [ $ln -gt $nb ] && continue
  echo "plan_batches=$nb plan_lanes=$ln stagger_bounce=$sb"; run --opt plan_batches=$nb --opt plan_lanes=$ln --opt stagger_bounce=$sb
done; done; done
done
