# on the GPU box: bash tools/dbg/tail_trace.sh "<bench options>"   -- per-kernel start / duration of the LAST launches of a bench run (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-.}; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/tt; timeout 240 rocprofv3 --kernel-trace -d /tmp/tt -o tt --output-format csv -- python $R/bench.py --no-cpu-baseline --no-roofline --no-configs $1 > /tmp/tt.log 2>&1
tail -1 /tmp/tt.log | cut -c1-200
python - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/tt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows)
last = [r for r in rows if int(r["Start_Timestamp"]) > t_end - 30_000_000]      # the last 30 ms
t0 = int(last[0]["Start_Timestamp"])
def short(n):
    m = re.search(r"k_trace<(\d+), (\w+), (\d+)>", n)
    if m: return "k_trace<%s>" % m.group(3)
    return re.sub(r"\(.*", "", n).replace("tirt::", "")[:28]
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.3f ms  +%8.3f ms  q%-3s %s  grid %s" % ((s - t0) / 1e6, (e - s) / 1e6, r.get("Queue_Id", "?"), short(r["Kernel_Name"]), r.get("Grid_Size", "?")))
PY
