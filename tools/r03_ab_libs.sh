# A/B of library builds: bash tools/r03_ab_libs.sh <tag> "<extra bench args>" lib1 lib2 ...
R=$GRAFT_REPO_ROOT; T=$1; X="$2"; shift 2
cd $R
for L in "$@"; do
  for o in "" "--opt trace_grid_alone=512" ; do
    echo -n "$L $o $X : "
    TIRT_LIB_PATH=$R/$L timeout 300 python bench.py --no-cpu-baseline --no-roofline $o $X 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"
  done
done 2>&1 | tee gpurun_out/${T}_ab_libs.log
