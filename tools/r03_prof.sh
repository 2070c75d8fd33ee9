# one-lane kernel stats + SQ counters for k_trace (round 3)
R=$GRAFT_REPO_ROOT; T=${1:-r03b}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats1lane -- $B > $R/gpurun_out/${T}_stats1lane.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_sq -- $B > $R/gpurun_out/${T}_sq.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/${T}_sq2 -- $B > $R/gpurun_out/${T}_sq2.log 2>&1
cd $R
python - <<PY
import csv,glob,collections
for sub in ("sq","sq2"):
    fs=glob.glob("gpurun_out/${T}_%s/**/*counter_collection.csv"%sub, recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
    for r in csv.DictReader(open(max(fs,key=lambda f: __import__('os').path.getsize(f)))):
        k=r["Kernel_Name"].split("(")[0].replace("void tirt::","")
        if not k.startswith("k_trace") and not k.startswith("k_shade"): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
    for k in agg: print(sub,k,len(n[k]),"launches", "%.3f ms avg"%(dur[k]/len(n[k])/1e6), {c: "%.4g"%(v/len(n[k])) for c,v in agg[k].items()})
PY
f=$(ls gpurun_out/${T}_stats1lane/*/*kernel_stats.csv | head -1); head -12 $f
