import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import oracle_api as oa
from common import tiny_scene, duplicate_code_scene
from test_gpu_trace import _grazing_rays
from ti_raytrace_amd import scenes, _native
for name, make, n, seed in (("cornell", lambda: scenes.cornell_box(48, 48, 4, device_id=0), 900, 17), ("tiny", lambda: tiny_scene(3000, seed=31, W=48, H=48, spread=0.08, device_id=0), 900, 17), ("dup", lambda: duplicate_code_scene(W=48, H=48, device_id=0), 900, 17),
                            ("100k", lambda: scenes.synthetic(64, 64, 4, device_id=0), 1500, 23)):
    ex = make(); ex.scene.setup_data_cpu()
    rays = _grazing_rays(ex, n, seed)
    ex = make(); ex.build_scene()
    if not ex.cam.view_inv_np.any(): ex.frame_camera(0.8)
    o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
    want, wprim, _ = o.closest_hit(rays)
    got, gprim, _ = ex.scene.ctx.trace_closest(rays, 64, 0)
    gote, gprime, _ = ex.scene.ctx.trace_closest(rays, 64, 1)
    print("exhaustive-mode mismatches vs oracle:", int((gprime != wprim).sum()))
    bad = np.flatnonzero(gprim != wprim)
    print(name, "rays", rays.shape[0], "bad", bad.size, "groups", np.bincount(bad // n, minlength=rays.shape[0] // n))
    for i in bad[:12]:
        print("  ray", i, "group", i // n, "o", rays[i, :3], "d", rays[i, 3:], "want prim", wprim[i], "t", want[i, 0], "got prim", gprim[i], "t", got[i, 0])
    if name == "100k":
        for i in (4615, 5182, 10897):
            print("  watch", i, "want", wprim[i], want[i, 0], "got", gprim[i], got[i, 0], "exh", gprime[i], gote[i, 0])
