# texture-address unit busy per kernel of the PT wavefront (one lane): bash tools/pmc_ta_pt.sh
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/ta_pt -- $B > $R/gpurun_out/ta_pt.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,os
fs=glob.glob("gpurun_out/ta_pt/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
for r in csv.DictReader(open(max(fs,key=os.path.getsize))):
    k=r["Kernel_Name"].split("(")[0].replace("void tirt::","").replace("tirt::","")[:24]
    if not (k.startswith("k_trace") or k.startswith("k_shade") or k.startswith("k_gen") or k.startswith("k_film")): continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
for k in sorted(agg, key=lambda k:-dur[k]):
    a=agg[k]; cyc=a["GRBM_GUI_ACTIVE"]/8.0
    print("%-24s %7.2f ms  TA busy %.2f  L2 read req %.3e write req %.3e  vmem rd/wave %.1f wr/wave %.1f" % (k, dur[k]/1e6, a["TA_TA_BUSY_sum"]/(cyc*256), a["TCP_TCC_READ_REQ_sum"], a["TCP_TCC_WRITE_REQ_sum"], a["SQ_INSTS_VMEM_RD"]/max(a["SQ_WAVES"],1), a["SQ_INSTS_VMEM_WR"]/max(a["SQ_WAVES"],1)))
PY
