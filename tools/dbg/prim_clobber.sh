#!/bin/bash
# ROCm 7.2 (roc-7.2.0, clang 22.0.0git 7b800a19) register-allocation problem behind the `prim` reload in the hit verification -- reproducer, as far as it could be reduced.
#   bash tools/dbg/prim_clobber.sh            (needs hipcc and this repository's git history; no GPU)
#
# The source that shows it is this repository's own k_trace as of commit d3df674^ (round 5) compiled with -DTR_NO_PRIM_RELOAD: the primitive id `prim` comes with the
# leaf's primitive record (its last word) and is used AFTER the ancestor walk of the hit verification when the candidate is still accepted.  The script checks that tree
# out of git into a temporary directory, compiles tirt_render.hip for gfx950 and prints, for k_trace<ordered, closest>: (1) the load of the record, (2) the loads inside
# the walk's loop that write the SAME register, (3) the read of the id after the walk.  With ROCm 7.2's hipcc (-O3) the record's last quad and the walk's compact-node
# rows share v[8:11]: the lanes that go through the walk lose the id and keep whatever the last row left in v11 -- 156 of 15 000 box-grazing rays on the Cornell box
# (tests/test_gpu_trace.py::test_quantised_nodes_on_grazing_rays failed on the device with that build).
#
# What a smaller reproducer would need, and why there is none (round 6): the collision is a decision of the register allocator over the whole 900-instruction kernel at
# 80 VGPRs (launch_bounds 256 x 6 waves).  It does not survive changes to the surrounding code: the SAME leaf step without the reload, in round 6's code shape
# (trace_leaf_step inlined from tirt_internal.h), gets v[12:15] / v[4:5] / v18 for the walk's loads and keeps the id in v11 untouched (second part of this script) --
# so neither a cut-down kernel nor today's source shows it, only the historical translation unit does.  The product does not depend on the outcome either way: the id
# is read again after the walk (trace_leaf_step), nothing of the record has to survive it, and tests/test_gpu_trace.py::test_hits_accepted_through_the_ancestor_walk
# compares the whole hit record of exactly those rays with the oracle.  __graft_entry__.build() still refuses another compiler unless TIRT_ALLOW_UNVALIDATED=1.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d ${TMPDIR:-/tmp}/prim_clobber.XXXXXX)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -S --cuda-device-only"
/opt/rocm/bin/hipcc --version | grep -E "HIP version|clang version" || true
analyse() {
python3 - "$1" <<'PY'
import re, sys
L = open(sys.argv[1]).read().split('\n')
a = next(i for i, l in enumerate(L) if l.startswith('_ZN4tirt7k_traceILi0ELb0ELi0EEEvNS_9TraceArgsE:'))
b = next(i for i in range(a, len(L)) if 's_endpgm' in L[i])
K = L[a:b]
rec = next(i for i, l in enumerate(K) if re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off offset:32', l))
m = re.search(r'v\[(\d+):(\d+)\]', K[rec]); lo, hi = int(m.group(1)), int(m.group(2))
print("(1) primitive record, last quad (its .w is the primitive id) -> v[%d:%d]:" % (lo, hi)); print("   line %d: %s" % (rec, K[rec].strip()))
print("(2) loads inside the ancestor walk (inner loop, Depth=2, behind the record's load) that write v%d:" % hi)
found = []
depth2 = False
for i in range(rec + 1, len(K)):
    if K[i].startswith('.LBB'):
        depth2 = 'Depth=2' in K[i] or (i + 1 < len(K) and 'Depth=2' in K[i + 1])
    mm = re.search(r'global_load_dword(x\d)? v\[?(\d+)(?::(\d+))?\]?,', K[i])
    if mm and depth2:
        l0 = int(mm.group(2)); h0 = int(mm.group(3) or l0)
        if l0 <= hi <= h0: found.append(i); print("   line %d: %s" % (i, K[i].strip()))
        else: print("   (line %d: %s -- another register)" % (i, K[i].strip()))
if not found: print("   none: in this translation unit the id keeps a register of its own"); sys.exit(0)
print("(3) reads of v%d after the walk, before anything redefines it:" % hi)
n = 0
for i in range(found[-1] + 1, len(K)):
    t = K[i].strip()
    if re.match(r'v_mov_b32_e32 v\d+, v%d$' % hi, t):
        print("   line %d: %s    <- becomes the accepted hit's primitive id" % (i, t)); n += 1
        if n == 2: break
PY
}
echo "== the historical translation unit (git d3df674^, -DTR_NO_PRIM_RELOAD) =="
mkdir -p $T/old/ti_raytrace_amd/csrc $T/old/include
for f in tirt_render.hip tirt_internal.h tirt_device.h tirt_math.h tirt_spectral.h; do git -C $R show d3df674^:ti_raytrace_amd/csrc/$f > $T/old/ti_raytrace_amd/csrc/$f; done
git -C $R show d3df674^:include/tirt.h > $T/old/include/tirt.h
(cd $T/old/ti_raytrace_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -DTIRT_EXPERIMENTS -DTR_NO_PRIM_RELOAD -o $T/old.s tirt_render.hip 2>/dev/null)
analyse $T/old.s
echo "== today's source with the reload line removed =="
mkdir -p $T/new/ti_raytrace_amd/csrc $T/new/include
cp $R/ti_raytrace_amd/csrc/*.h $R/ti_raytrace_amd/csrc/tirt_render.hip $T/new/ti_raytrace_amd/csrc/; cp $R/include/tirt.h $T/new/include/
grep -v "prim = (int)b.compact\[(size_t)leaf \* CPN_VEC + 1\];" $R/ti_raytrace_amd/csrc/tirt_internal.h > $T/new/ti_raytrace_amd/csrc/tirt_internal.h
(cd $T/new/ti_raytrace_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -o $T/new.s tirt_render.hip 2>/dev/null)
analyse $T/new.s
rm -rf $T
