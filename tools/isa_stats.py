"""Instruction counts per kernel of a translation unit's gfx950 ISA (hipcc -S --cuda-device-only): VALU / SALU / VMEM / LDS / total, VGPRs, scratch.
   python tools/isa_stats.py ti_raytrace_amd/csrc/tirt_render.hip [kernel-name-substring ...]   (prints one line per kernel; used to show that a refactoring left the ISA alone)"""
import re, subprocess, sys, os, tempfile
src = sys.argv[1]; keys = sys.argv[2:]
d = os.path.dirname(os.path.abspath(src))
out = tempfile.mktemp(suffix=".s", dir="/tmp")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize".split() + os.environ.get("EXTRA", "").split()
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", out, os.path.abspath(src)], cwd=d)
txt = open(out).read().split("\n"); os.unlink(out)
cur = None; stats = {}
for l in txt:
    m = re.match(r"^(_Z\w+):", l)
    if m: cur = m.group(1); stats[cur] = {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "total": 0, "hash": 0}; continue
    if cur is None: continue
    t = l.strip()
    if t.startswith(".amdhsa_next_free_vgpr"): stats[cur]["vgpr_alloc"] = int(t.split()[1])
    if t.startswith("; NumVgprs:"): stats[cur]["vgprs"] = int(t.split()[2])
    if t.startswith("; ScratchSize:"): stats[cur]["scratch"] = int(t.split()[2])
    if t.startswith("; codeLenInByte"): stats[cur]["bytes"] = int(t.split()[-1])
    if not t or t[0] in ";." or t.endswith(":"): continue
    op = t.split()[0]
    s = stats[cur]; s["total"] += 1
    s["hash"] = (s["hash"] * 1000003 + hash(re.sub(r"\s+", " ", t.split(";")[0])) ) & 0xffffffffffff      # (order-sensitive digest of the instruction text)
    if op.startswith("v_"): s["valu"] += 1
    elif op.startswith("s_"): s["salu"] += 1
    elif op.startswith("ds_"): s["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): s["vmem"] += 1
import hashlib
for k, s in stats.items():
    if s["total"] == 0 or (keys and not any(x in k for x in keys)): continue
    dem = subprocess.run(["c++filt", k], stdout=subprocess.PIPE).stdout.decode().strip().split("(")[0].replace("void tirt::", "")
    print("%-44s VALU %5d SALU %5d VMEM %4d LDS %4d total %6d bytes %6s VGPRs %4s scratch %3s" % (dem[:44], s["valu"], s["salu"], s["vmem"], s["lds"], s["total"], s.get("bytes"), s.get("vgprs"), s.get("scratch")))
