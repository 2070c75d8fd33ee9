"""Prints per-kernel sums of the counters collected by tools/pmc_mem.sh, normalised by kernel time."""
import collections, csv, glob, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "mem"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
for sub in "abcdefg":
    d = os.path.join(root, "%s_%s" % (tag, sub))
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not cc or not kt:
        continue
    dur = collections.defaultdict(float); nl = collections.defaultdict(int)
    for r in csv.DictReader(open(max(kt, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tirt::", "")
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; nl[k] += 1
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(max(cc, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tirt::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        if not (k.startswith("k_trace") or k.startswith("k_shade")):
            continue
        t = dur[k]
        print("%s pass %s: %d launches %.2f ms" % (k, sub, nl[k], t * 1e3))
        for c, x in sorted(v.items()):
            print("    %-44s %.4g   per-ns %.4g" % (c, x, x / (t * 1e9)))
