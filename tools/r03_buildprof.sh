R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r03_buildprof -- python $R/tools/build_prof.py 100000 > $R/gpurun_out/r03_buildprof.log 2>&1
cd $R; python - <<'PY'
import csv,glob
f=max(glob.glob("gpurun_out/r03_buildprof/**/*kernel_trace.csv",recursive=True), key=lambda p: __import__('os').path.getsize(p))
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find the last k_morton -> treat as the start of the last build
idx=[i for i,r in enumerate(rows) if "k_morton" in r["Kernel_Name"]]
i0=idx[-1]
t0=int(rows[i0]["Start_Timestamp"]); prev_end=t0
for r in rows[i0:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    n=r["Kernel_Name"].split("(")[0].replace("void tirt::","").replace("tirt::","")[:28]
    print("%8.1f us  dur %6.1f  gap %6.1f  %s  grid %s" % ((s-t0)/1e3,(e-s)/1e3,(s-prev_end)/1e3,n,r["Grid_Size_X"]))
    prev_end=e
PY
