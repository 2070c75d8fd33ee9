R=${GRAFT_REPO_ROOT:-.}
for V in "$@"; do
  make -s -C $R/ti_raytrace_amd/csrc clean; make -s -C $R/ti_raytrace_amd/csrc EXTRA="$V" 2>&1 | grep -E "error"
  timeout 300 python $R/tools/run_configs.py 5 2>&1 | grep bdpt_512 | cut -c1-110 | sed "s/^/$V /"
done
