# SQ counters of k_pvb_cand / k_pvb_cand_refill on the one-lane command: bash tools/r06_pvb_pmc.sh <tag> [extra --opt ...]
R=$GRAFT_REPO_ROOT; T=${1:-r06h}; shift; cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
B1="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432 --opt primary_beams_refill=$v $@"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_sq$v -- $B1 > /dev/null 2>&1
python - <<PY
import csv,glob,collections,os
fs=glob.glob("$R/gpurun_out/${T}_sq$v/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
for r in csv.DictReader(open(max(fs,key=os.path.getsize))):
    k=r["Kernel_Name"].split("(")[0].replace("void tirt::","").replace("tirt::","")
    if not k.startswith("k_pvb_cand"): continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for k,a in agg.items():
    L=len(n[k]); cyc=a["GRBM_GUI_ACTIVE"]/8.0
    print(k, "launches",L,"avg_ms",round(dur[k]/L/1e6,4),"VALU wave-insts per launch %.1f M"%(a["SQ_INSTS_VALU"]/L/1e6),"issue_busy",round(a["SQ_ACTIVE_INST_VALU"]*4/(cyc*1024),3),
          "lane_util",round(a["SQ_THREAD_CYCLES_VALU"]/(a["SQ_ACTIVE_INST_VALU"]*64),3),"wait_any/wave_cycles",round(a["SQ_WAIT_ANY"]/a["SQ_WAVE_CYCLES"],3),
          "VMEM_RD per launch %.1f M"%(a["SQ_INSTS_VMEM_RD"]/L/1e6),"SALU %.1f M"%(a["SQ_INSTS_SALU"]/L/1e6))
PY
rm -rf $R/gpurun_out/${T}_sq$v
done
