#!/bin/bash
# The register-allocation problem behind the `prim` reload in k_trace's hit verification (tirt_render.hip), shown on the ISA.
#   bash tools/dbg/prim_clobber.sh            (needs hipcc; no GPU)
# Compiles tirt_render.hip for gfx950 with -DTR_NO_PRIM_RELOAD -- the source as it was before the work-around: the primitive id `prim` comes
# with the leaf's primitive record (its last word) and is used AFTER the ancestor walk of the hit verification when the candidate is still
# accepted -- and prints, for k_trace<ordered, closest>: (1) the load of the record, (2) the loads inside the walk's loop, (3) the read of the
# id after the walk.  With ROCm 7.2's hipcc (-O3) the record's last quad and the walk's compact-node rows share v[8:11]: the lanes that go
# through the walk (candidates whose leaf box fails the exact `slabs` test) lose the id and keep whatever the last row left in v11 --
# 156 of 15 000 box-grazing rays on the Cornell box, tests/test_gpu_trace.py::test_quantised_nodes_on_grazing_rays.
# Since round 4 the id is read again after the walk (from the leaf's compact row), so nothing of the record has to survive it.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/prim_clobber.s
/opt/rocm/bin/hipcc --version | grep -E "HIP version|clang version" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DTR_NO_PRIM_RELOAD \
    -S --cuda-device-only -o $OUT $R/ti_raytrace_amd/csrc/tirt_render.hip 2>/dev/null
python3 - $OUT <<'PY'
import re, sys
L = open(sys.argv[1]).read().split('\n')
a = next(i for i, l in enumerate(L) if l.startswith('_ZN4tirt7k_traceILi0ELb0ELi0EEEvNS_9TraceArgsE:'))
b = next(i for i in range(a, len(L)) if 's_endpgm' in L[i])
K = L[a:b]
rec = next(i for i, l in enumerate(K) if re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off offset:32', l))
m = re.search(r'v\[(\d+):(\d+)\]', K[rec]); lo, hi = int(m.group(1)), int(m.group(2))
print("(1) primitive record, last quad (its .w is the primitive id) -> v[%d:%d]:" % (lo, hi)); print("   line %d: %s" % (rec, K[rec].strip()))
print("(2) loads inside the ancestor walk (inner loop, Depth=2) that write v%d:" % hi)
found = []
depth2 = False
for i in range(rec + 1, len(K)):
    if K[i].startswith('.LBB'):
        depth2 = 'Depth=2' in K[i] or (i + 1 < len(K) and 'Depth=2' in K[i + 1])
    mm = re.search(r'global_load_dword(x\d)? v\[?(\d+)(?::(\d+))?\]?,', K[i])
    if mm and depth2:
        l0 = int(mm.group(2)); h0 = int(mm.group(3) or l0)
        if l0 <= hi <= h0: found.append(i); print("   line %d: %s" % (i, K[i].strip()))
if not found: print("   none: this compiler keeps the id in a register of its own"); sys.exit(0)
print("(3) reads of v%d after the walk, before anything redefines it:" % hi)
n = 0
for i in range(found[-1] + 1, len(K)):
    t = K[i].strip()
    if re.match(r'v_mov_b32_e32 v\d+, v%d$' % hi, t):
        print("   line %d: %s    <- becomes the accepted hit's primitive id" % (i, t)); n += 1
        if n == 2: break
PY
