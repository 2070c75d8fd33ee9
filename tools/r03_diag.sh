R=$GRAFT_REPO_ROOT; cd $R
for L in "$@"; do
TIRT_LIB_PATH=$R/$L timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', d['value'], r['kernel_ms'], r['wave_diag_ordered'], r['node_visits_per_ray'], r['prim_tests_per_ray'], r['lds_node_visits_per_ray'])"
done
