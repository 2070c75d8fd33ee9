# SQ counters of k_trace for a given library: bash tools/r03_prof2.sh <tag> <lib>
R=$GRAFT_REPO_ROOT; T=$1; export TIRT_LIB_PATH=$R/$2
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/${T}_sq -- $B > $R/gpurun_out/${T}_sq.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,os
fs=glob.glob("gpurun_out/${T}_sq/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
for r in csv.DictReader(open(max(fs,key=os.path.getsize))):
    k=r["Kernel_Name"].split("(")[0].replace("void tirt::","")
    if not k.startswith("k_trace"): continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for k in agg: print("$2",k,len(n[k]),"launches", "%.3f ms avg"%(dur[k]/len(n[k])/1e6), {c: "%.4g"%(v/len(n[k])) for c,v in agg[k].items()})
PY
