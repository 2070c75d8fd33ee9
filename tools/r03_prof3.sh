# SQ + memory counters of every kernel of the PT wavefront (one lane): bash tools/r03_prof3.sh <tag>
R=$GRAFT_REPO_ROOT; T=$1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${T}_sq -- $B > $R/gpurun_out/${T}_sq.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $R/gpurun_out/${T}_mem -- $B > $R/gpurun_out/${T}_mem.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,os
for tag in ("sq","mem"):
    fs=glob.glob("gpurun_out/${T}_%s/**/*counter_collection.csv"%tag, recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
    for r in csv.DictReader(open(max(fs,key=os.path.getsize))):
        k=r["Kernel_Name"].split("(")[0].replace("void tirt::","")[:40]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
    for k in sorted(agg, key=lambda k:-dur[k])[:8]: print(tag,k,len(n[k]),"launches", "total %.3f ms"%(dur[k]/1e6), {c: "%.4g"%(v) for c,v in agg[k].items()})
PY
