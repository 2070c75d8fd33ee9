cd $GRAFT_REPO_ROOT
for v in "" pv_x1.25 pv_x1.5 pv_x2.0 pv_x4.0; do
  echo "== ${v:-x1.0001}"
  if [ -n "$v" ]; then export TIRT_LIB_PATH=$GRAFT_REPO_ROOT/ab_libs/$v.so; else unset TIRT_LIB_PATH; fi
  for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 20 --warmup 5 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['primary_beams']; print('  headline', d['value'], p['pixels_with_list'], p['leaves_per_listed_pixel'], p['share_to_k_trace'])"; done
done
