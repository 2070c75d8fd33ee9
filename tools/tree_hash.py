"""hash of the traversal tree (and build time) for a few scenes: python tools/tree_hash.py  (TIRT_LIB_PATH selects the library)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from ti_raytrace_amd import scenes
from common import tiny_scene, duplicate_code_scene
for name, make in (("cornell", lambda: scenes.cornell_box(32, 32, 4, device_id=0)), ("teapot", lambda: scenes.single_model(32, 32, 4, device_id=0)),
                   ("veach", lambda: scenes.veach_bdpt(32, 32, 4, device_id=0, integrator="pt")), ("synthetic100k", lambda: scenes.synthetic(32, 32, 4, device_id=0)),
                   ("tiny3000", lambda: tiny_scene(3000, seed=5, W=32, H=32, spread=0.08, device_id=0)), ("tiny2", lambda: tiny_scene(2, seed=3, W=16, H=16, spread=0.5, device_id=0)),
                   ("points5000", lambda: tiny_scene(5000, seed=6, W=16, H=16, spread=0.0001, device_id=0)), ("dup", lambda: duplicate_code_scene(W=16, H=16, device_id=0)),
                   ("synthetic1M", lambda: scenes.synthetic(32, 32, 4, ntri=1000000, spread=0.012, device_id=0))):
    ex = make(); ex.build_scene(); ctx = ex.scene.ctx
    rows = ctx.traversal_tree_download(ex.scene.primitive_count)
    ms = []
    for _ in range(4):
        ctx.lbvh_build(); ms.append(ctx.stats()["ms_build"])
    print("%-14s %8d prims  tree %s  build %.3f ms" % (name, ex.scene.primitive_count, hashlib.blake2b(rows.tobytes(), digest_size=8).hexdigest(), min(ms[1:])))
