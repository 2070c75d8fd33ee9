# HBM bytes of the BDPT kernels (config 5, one lane): bash tools/pmc_bdpt4.sh   (FETCH_SIZE / WRITE_SIZE in units of 32 B on gfx950 after the guide's correction: reported as kilobytes x ...; printed raw and as GB at 32 B)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python $R/tools/bdpt_bench.py 64 512 overlap_lanes=1"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/bd4_a -- $B > $R/gpurun_out/bd4_a.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/bd4_b -- $B > $R/gpurun_out/bd4_b.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum --output-format csv -d $R/gpurun_out/bd4_c -- $B > $R/gpurun_out/bd4_c.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,os
for tag in ("a","b","c"):
    fs=glob.glob("gpurun_out/bd4_%s/**/*counter_collection.csv"%tag, recursive=True)
    if not fs: print("pass", tag, "gave no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
    for r in csv.DictReader(open(max(fs,key=os.path.getsize))):
        k=r["Kernel_Name"].split("(")[0].replace("void tirt::","").replace("tirt::","")[:28]
        if not k.startswith("k_bd") and not k.startswith("k_trace"): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
    for k in sorted(agg, key=lambda k:-dur[k]):
        print("%-28s %7.2f ms (%d launches) " % (k, dur[k]/1e6, len(n[k])) + " ".join("%s %.4e" % (c, v) for c, v in sorted(agg[k].items())))
PY
