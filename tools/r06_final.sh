# round-6 measurement set on the GPU box: bash tools/r06_final.sh <tag>  -> gpurun_out/<tag>_*
R=$GRAFT_REPO_ROOT; T=${1:-r06z}; cd $R
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/${T}_pytest_gpu.log 2>&1; tail -1 gpurun_out/${T}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${T}_bench_default.json.log 2> gpurun_out/${T}_bench_default.err; tail -c 400 gpurun_out/${T}_bench_default.json.log; echo
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver_command_steps20_warmup5.json.log 2> /dev/null; tail -c 200 gpurun_out/${T}_bench_driver_command_steps20_warmup5.json.log; echo
TIRT_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-configs --no-traffic > gpurun_out/${T}_bench_force_dist_one_rank_rccl.json.log 2>&1
for n in 1 2 4 8; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world $n 2>/dev/null | tail -1; done > gpurun_out/${T}_emulated_rank_scaling.log
for s in 4 8 16 32; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --emulate-world 8 --steps $s 2>/dev/null | tail -1; done > gpurun_out/${T}_emulated_8_ranks_steps_4_8_16_32.log
for n in 1 2 4 8; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 20 --warmup 5 --emulate-world $n 2>/dev/null | tail -1; done > gpurun_out/${T}_emulated_rank_scaling_driver_command_steps20.log
timeout 300 python tools/timeline.py 0 1 4 8 > gpurun_out/${T}_wave_timeline.txt 2>&1
timeout 200 python tools/dbg/long_rays.py > gpurun_out/${T}_long_rays.txt 2>&1
timeout 200 python tools/dbg/ray_kinds.py > gpurun_out/${T}_ray_kinds.txt 2>&1
for i in 1 2; do python tools/bdpt_bench.py 64 512; done > gpurun_out/${T}_bdpt_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
B1="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --opt overlap_lanes=1 --opt batch_paths=33554432 --opt merge_paths=33554432"
B4="python $R/bench.py --no-cpu-baseline --no-roofline"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats1lane -- $B1 > $R/gpurun_out/${T}_stats1lane.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats4lanes -- $B4 > $R/gpurun_out/${T}_stats4lanes.log 2>&1
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_sq -- $B1 > $R/gpurun_out/${T}_sq.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_bdpt_stats1lane -- python $R/tools/bdpt_bench.py 64 512 overlap_lanes=1 > $R/gpurun_out/${T}_bdpt_stats1lane.log 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${T}_bdpt_sq -- python $R/tools/bdpt_bench.py 64 512 overlap_lanes=1 > $R/gpurun_out/${T}_bdpt_sq.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,os,json
out={}
for tag in ("sq","bdpt_sq"):
    fs=glob.glob("gpurun_out/${T}_%s/**/*counter_collection.csv"%tag, recursive=True)
    if not fs: continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set); dur=collections.defaultdict(float)
    for r in csv.DictReader(open(max(fs,key=os.path.getsize))):
        k=r["Kernel_Name"].split("(")[0].replace("void tirt::","").replace("tirt::","")
        if k.startswith("k_sah") or k.startswith("k_wide") or k.startswith("__amd") or k.startswith("k_morton") or k.startswith("k_radix"): continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Dispatch_Id"] not in n[k]: n[k].add(r["Dispatch_Id"]); dur[k]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
    o={}
    for k in sorted(agg, key=lambda k:-agg[k]["SQ_INSTS_VALU"])[:8]:
        a=agg[k]; cyc=a["GRBM_GUI_ACTIVE"]/8.0
        o[k]={"launches":len(n[k]),"total_ms":round(dur[k]/1e6,3),"avg_ms":round(dur[k]/len(n[k])/1e6,4),"SQ_INSTS_VALU":a["SQ_INSTS_VALU"],
              "valu_issue_busy":round(a["SQ_ACTIVE_INST_VALU"]*4/(cyc*1024),4) if cyc else None,
              "lane_util":round(a["SQ_THREAD_CYCLES_VALU"]/(a["SQ_ACTIVE_INST_VALU"]*64),4) if a.get("SQ_THREAD_CYCLES_VALU") else None}
    out[tag]=o
json.dump(out, open("gpurun_out/${T}_pmc_summary.json","w"), indent=1)
print(json.dumps(out)[:1500])
PY
echo done
