# TA / L2 / wave-occupancy counters of the BDPT kernels (config 5, one lane): bash tools/pmc_bdpt2.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python $R/tools/bdpt_bench.py 64 512 overlap_lanes=1"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $R/gpurun_out/bd2_a -- $B > $R/gpurun_out/bd2_a.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/bd2_b -- $B > $R/gpurun_out/bd2_b.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,os
for tag in ("a","b"):
    fs=glob.glob("gpurun_out/bd2_%s/**/*counter_collection.csv"%tag, recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
    for r in csv.DictReader(open(max(fs,key=os.path.getsize))):
        k=r["Kernel_Name"].split("(")[0].replace("void tirt::","").replace("tirt::","")[:28]
        if not k.startswith("k_bd") and not k.startswith("k_trace"): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
    for k in sorted(agg, key=lambda k:-dur[k]):
        a=agg[k]; cyc=a["GRBM_GUI_ACTIVE"]/8.0
        if tag=="a": print("%-28s %7.2f ms  TA busy %.2f  L2 read req %.3e (%.1f GB at 64B..128B) write req %.3e" % (k, dur[k]/1e6, a["TA_TA_BUSY_sum"]/(cyc*256), a["TCP_TCC_READ_REQ_sum"], a["TCP_TCC_READ_REQ_sum"]*64/1e9, a["TCP_TCC_WRITE_REQ_sum"]))
        else: print("%-28s waves %.3e  mean wave cycles %.0f  occupancy (wave-cycles / (cyc*1024*... )) %.2f waves/SIMD  vmem rd %.3e wr %.3e" % (k, a["SQ_WAVES"], a["SQ_WAVE_CYCLES"]*4/max(a["SQ_WAVES"],1), a["SQ_WAVE_CYCLES"]*4/(cyc*1024), a["SQ_INSTS_VMEM_RD"], a["SQ_INSTS_VMEM_WR"]))
PY
