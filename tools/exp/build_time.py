"""LBVH + traversal-tree build times (HIP events, third build of the same scene) for a synthetic scene of N triangles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ti_raytrace_amd import scenes
for ntri in [int(x) for x in sys.argv[1:]] or [100000]:
    for tree in (0, 1):
        ex = scenes.synthetic(64, 64, 4, ntri=ntri, device_id=0)
        ctx = ex.scene.ctx
        ctx.set_option("traversal_tree", tree)
        ex.build_scene()
        ms = []
        for rep in range(3):
            ctx.lbvh_build(); ms.append(ctx.stats()["ms_build"])
        print("ntri %d tree=%d build ms %s" % (ntri, tree, ["%.3f" % m for m in ms]))
        del ex, ctx
