"""Joins the plain output of tools/micro/hbm_gather with its rocprofv3 --pmc passes (counter_collection.csv under the given directories): per kernel the bytes it is
known to need against what FETCH_SIZE / TCC_EA0_RDREQ / TCC_MISS report -- the calibration of bench.py's `fetch_correction_calibrated_on: gather`.
   python tools/hbm_gather_summary.py plain.txt dir_fetch dir_ea [dir_write] > profiles/r06_micro_hbm_gather.txt"""
import csv, glob, os, re, sys
plain = open(sys.argv[1]).read().splitlines()
ctr = {}
for d in sys.argv[2:]:
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(max(fs, key=os.path.getsize))):
        m = re.search(r"k_hbm<(\d+), ?(\d+)>", r["Kernel_Name"])
        if not m: continue
        k = (int(m.group(1)), int(m.group(2)))
        e = ctr.setdefault(k, {})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e.setdefault("_ids_" + r["Counter_Name"], set()).add(r["Dispatch_Id"])
print(plain[0])
print("%-9s %9s %14s %12s | %10s %9s | %14s %9s %9s | %9s %9s" % ("kind", "ws MB", "known MB", "GB/s known", "FETCH MB", "x known", "EA_RDREQ", "B/req", "32B share", "TCC_MISS", "hit rate"))
for l in plain[1:]:
    m = re.match(r"(\w+)\s+ws\s+(\d+) MB\s+kernel k_hbm<(\d),(\d)>.*known_bytes (\d+)\s+best ([\d.]+) ms\s+([\d.]+) GB/s", l)
    if not m: continue
    kind, ws, K, W, known, ms, gbs = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)), float(m.group(6)), float(m.group(7))
    e = ctr.get((K, W), {})
    def per_launch(c):
        n = len(e.get("_ids_" + c, ())) or 1
        return e.get(c, 0.0) / n if c in e else None
    f = per_launch("FETCH_SIZE"); ea = per_launch("TCC_EA0_RDREQ_sum"); ea32 = per_launch("TCC_EA0_RDREQ_32B_sum"); miss = per_launch("TCC_MISS_sum"); hit = per_launch("TCC_HIT_sum")
    print("%-9s %9d %14.1f %12.1f | %10s %9s | %14s %9s %9s | %9s %9s" % (kind, ws, known / 1e6, gbs,
          "%.1f" % (f * 1024 / 1e6) if f is not None else "-", "%.3f" % (f * 1024 / known) if f is not None else "-",
          "%.0f" % ea if ea is not None else "-", "%.1f" % (known / ea) if ea else "-", "%.3f" % (ea32 / ea) if ea and ea32 is not None else "-",
          "%.0f" % miss if miss is not None else "-", "%.3f" % (hit / (hit + miss)) if miss is not None and hit is not None and hit + miss > 0 else "-"))
