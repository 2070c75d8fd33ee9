# on the GPU box: bash tools/dbg/sq_ab.sh <lib.so | ""> <tag>  -- SQ counters (serialised dispatches) of one 32-frame step: per kernel calls, average duration, VALU wave-instructions, lane utilisation
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${2:-sq}
[ -n "$1" ] && export TIRT_LIB_PATH=$1
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-configs --opt batch_paths=33554432 --opt merge_paths=33554432"
rm -rf /tmp/$T; timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/$T -- $B > /tmp/$T.log 2>&1
python - <<PY
import csv, glob, collections, re
f = glob.glob("/tmp/$T/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]; m = re.search(r"k_trace<(\d+), (\w+), (\d+)>", n)
    k = ("k_trace<%s>" % m.group(3)) if m else re.sub(r"\(.*", "", n).replace("void ", "").replace("tirt::", "")[:24]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
print("$T")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:6]:
    print("  %-22s calls %4d  VALU wave-insts %.4g  lane util %.3f  busy cycles (GRBM) %.4g  wave cycles %.4g" % (k, len(calls[k]), v["SQ_INSTS_VALU"],
          v["SQ_THREAD_CYCLES_VALU"] / max(v["SQ_ACTIVE_INST_VALU"] * 64.0, 1), v["GRBM_GUI_ACTIVE"], v["SQ_WAVE_CYCLES"]))
PY
