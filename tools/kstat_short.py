"""one line per BDPT kernel of a rocprofv3 --kernel-trace --stats directory: python tools/kstat_short.py <dir>"""
import csv, glob, sys
fs = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not fs: print("no kernel_stats.csv under", sys.argv[1]); sys.exit(0)
out = []
for r in csv.DictReader(open(fs[0])):
    k = r["Name"].split("(")[0].replace("void tirt::", "").replace("tirt::", "")
    if k.startswith("k_bd") or k.startswith("k_trace"): out.append("%s x%s %.2f ms" % (k, r["Calls"], float(r["TotalDurationNs"]) / 1e6))
print("   " + " | ".join(out[:7]))
