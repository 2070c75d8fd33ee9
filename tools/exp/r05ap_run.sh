cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for dv in 1 2 4 8; do
export TIRT_PVB_GRID_DIV=$dv
B1="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432"
rm -rf /tmp/pv; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- $B1 > /tmp/pv.log 2>&1
echo "== div $dv"; python - <<PY
import csv,glob
f=glob.glob("/tmp/pv/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    n=r["Name"].replace("tirt::","").split("(")[0]
    if "k_trace<0, false, 0>" in n: print("   %-40s x%-4s avg %8.4f ms" % (n[:40], r["Calls"], float(r["AverageNs"])/1e6))
PY
for i in 1 2; do python $R/bench.py --no-cpu-baseline --no-roofline --no-configs --steps 20 --warmup 5 | tail -1 | cut -c90-170; done
done
