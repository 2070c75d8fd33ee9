# VALU issue of the BDPT kernels (config 5, one lane): bash tools/pmc_bdpt3.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python $R/tools/bdpt_bench.py 64 512 overlap_lanes=1"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/bd3_c -- $B > $R/gpurun_out/bd3_c.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/bd3_d -- $B > $R/gpurun_out/bd3_d.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,os
for tag in ("c","d"):
    fs=glob.glob("gpurun_out/bd3_%s/**/*counter_collection.csv"%tag, recursive=True)
    if not fs: print("pass", tag, "gave no counters"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
    for r in csv.DictReader(open(max(fs,key=os.path.getsize))):
        k=r["Kernel_Name"].split("(")[0].replace("void tirt::","").replace("tirt::","")[:28]
        if not k.startswith("k_bd") and not k.startswith("k_trace"): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
    for k in sorted(agg, key=lambda k:-dur[k]):
        a=agg[k]; cyc=a["GRBM_GUI_ACTIVE"]/8.0
        print("%-28s %7.2f ms " % (k, dur[k]/1e6) + " ".join("%s %.3e (per SIMD-cycle %.3f)" % (c, v, v/(cyc*1024)) for c, v in sorted(a.items()) if c != "GRBM_GUI_ACTIVE"))
PY
