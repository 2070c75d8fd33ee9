"""One-off robustness check: LBVH parity + a short render on a large synthetic scene (default 1 M triangles)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as oa
from ti_raytrace_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
W = H = 512
t = time.time(); ex = scenes.synthetic(W, H, 8, ntri=n, spread=0.012, device_id=0); ex.build_scene(); ex.scene.ctx.sync()
print("setup+GPU build wall %.2f s, GPU build %.3f ms" % (time.time() - t, ex.scene.ctx.stats()["ms_build"]))
o = oa.OracleScene(ex.scene, ex.cam); t = time.time(); assert o.lbvh_build() == n; print("oracle build %.2f s" % (time.time() - t))
om, ob, oc = o.lbvh_get(); gm, gb, gc = ex.scene.ctx.lbvh_download(n + 1)
print("sorted pairs equal:", np.array_equal(gm, om), " bvh_node equal:", np.array_equal(gb.view(np.uint32), ob.view(np.uint32)),
      " compact equal:", np.array_equal(gc.view(np.uint32), oc.view(np.uint32)), " duplicate codes:", int((np.diff(om[:, 0]) == 0).sum()))
ex.scene.ctx.stats_reset(); t = time.time(); ex.integrator.render_frames(8); ex.scene.ctx.sync(); dt = time.time() - t
st = ex.scene.ctx.stats()
print("render 512^2 x8: %.3f s, %.1f Mrays/s, overflow %d" % (dt, (st["rays_closest"] + st["rays_shadow"]) / dt / 1e6, st["stack_overflow"]))
rays = oa.camera_rays(ex.cam, W, H)[::37]
want, wprim, _ = o.closest_hit(rays); got, gprim, _ = ex.scene.ctx.trace_closest(rays, 64, 0)
print("closest-hit prims equal on %d rays:" % len(rays), np.array_equal(gprim, wprim), " t bits equal:", np.array_equal(got[:, 0].view(np.uint32), want[:, 0].view(np.uint32)))
