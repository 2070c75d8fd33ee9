"""Per-kernel VGPR / scratch / LDS / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage (stdin or a .hip file)."""
import re, subprocess, sys
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split()
src = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[3:] + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if pat in k: print("%-60s vgpr %3d scratch %4d lds %6d occ %d" % (k, v.get("VGPRs", -1), v.get("ScratchSize", -1), v.get("LDS Size", -1), v.get("Occupancy", -1)))
