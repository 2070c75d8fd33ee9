cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for pb in 0 1; do
B1="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432 --opt primary_beams=$pb"
rm -rf /tmp/pv$pb; timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv$pb -- $B1 > /tmp/pv$pb.log 2>&1
tail -1 /tmp/pv$pb.log | cut -c1-200
python - <<PY
import csv,glob
f=glob.glob("/tmp/pv$pb/**/*kernel_stats.csv",recursive=True)[0]
for r in [x for x in csv.DictReader(open(f)) if "pvb" in x["Name"] or "k_trace" in x["Name"]]: print("   %-40s x%-4s total %8.3f ms  avg %8.4f ms" % (r["Name"].replace("tirt::","").split("(")[0][:40], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
done
for pb in; do python $R/bench.py --no-cpu-baseline --no-roofline --no-configs --steps 20 --warmup 5 --opt primary_beams=$pb | tail -1 | cut -c90-220; done
