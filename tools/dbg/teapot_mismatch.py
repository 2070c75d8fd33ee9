"""Which rays differ between the ordered and the reference-order traversal on the Teapot's grazing rays (debug)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ti_raytrace_amd import scenes, _native
from test_gpu_trace import _grazing_rays
import oracle_api as oa
ex = scenes.single_model(32, 32, 4, device_id=0); ex.build_scene(); ctx = ex.scene.ctx
n = 1000000
rays = _grazing_rays(ex, max(n // 14, 100), 41).astype(np.float32)
for opt in ({}, {"traversal_tree": 0}):
    for k, v in opt.items(): ctx.set_option(k, v); 
    if opt: ctx.lbvh_build()
    a, ap, _ = ctx.trace_closest(rays, 64, 0)
    b, bp, _ = ctx.trace_closest(rays, 64, _native.TRAVERSE_EXHAUSTIVE)
    bad = np.nonzero((ap != bp) | (a[:, 0].view(np.uint32) != b[:, 0].view(np.uint32)))[0]
    print(opt, "mismatching rays:", len(bad))
    P = np.asarray(ex.scene.primitive_np).reshape(-1, 3) if hasattr(ex.scene, "primitive_np") else None
    for i in bad[:6]:
        print("  ray", i, rays[i], "ordered prim %d t %.9g | reference-order prim %d t %.9g" % (ap[i], a[i, 0], bp[i], b[i, 0]))
        if P is not None:
            for pid in (ap[i], bp[i]):
                if pid >= 0: print("     prim", pid, "type", P[pid, 0])
