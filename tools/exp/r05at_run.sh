cd $GRAFT_REPO_ROOT
for v in "" pv_y1.1 pv_y1.25 pv_y1.5 pv_y2.0; do
  echo "== ${v:-x1.0001}"
  if [ -n "$v" ]; then export TIRT_LIB_PATH=$GRAFT_REPO_ROOT/ab_libs/$v.so; else unset TIRT_LIB_PATH; fi
  for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 20 --warmup 5 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['primary_beams']; print('  headline', d['value'], p['pixels_with_list'], p['leaves_per_listed_pixel'], p['share_to_k_trace'])"; done
done
for v in "" pv_y1.25 pv_y1.5; do
  if [ -n "$v" ]; then export TIRT_LIB_PATH=$GRAFT_REPO_ROOT/ab_libs/$v.so; else unset TIRT_LIB_PATH; fi
  echo "== ${v:-x1.0001} other scenes"
  python bench.py --configs-only big_scene_4M_1024x1024_32spp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=list(d.values())[0]; print('  big', v['Mrays_per_s'], v['oracle_sample_identical'])"
  python bench.py --configs-only config2_teapot_1024x1024_64spp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=list(d.values())[0]; print('  teapot', v['Mrays_per_s'], v['oracle_sample_identical'])"
done
unset TIRT_LIB_PATH; python -m pytest tests/test_gpu_beams.py -m gpu -q 2>&1 | grep -E "passed|failed" | sed 's/^/  /'
