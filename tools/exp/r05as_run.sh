cd $GRAFT_REPO_ROOT
for v in "" pv_c12 pv_c48; do
  echo "== ${v:-c24}"
  if [ -n "$v" ]; then export TIRT_LIB_PATH=$GRAFT_REPO_ROOT/ab_libs/$v.so; else unset TIRT_LIB_PATH; fi
  for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 20 --warmup 5 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  headline', d['value'], d['primary_beams']['pixels_with_list'], d['primary_beams']['share_to_k_trace'])"; done
  TIRT_BENCH_CTX_OPTS= python bench.py --configs-only big_scene_4M_1024x1024_32spp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=list(d.values())[0]; print('  big', v['Mrays_per_s'])"
  python bench.py --configs-only config2_teapot_1024x1024_64spp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=list(d.values())[0]; print('  teapot', v['Mrays_per_s'])"
  python bench.py --configs-only config1_cornell_512x512_512spp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=list(d.values())[0]; print('  cornell', v['Mrays_per_s'])"
done
