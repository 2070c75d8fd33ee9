"""Builds the LBVH of an n-triangle synthetic scene once (for `rocprofv3 --kernel-trace --stats -- python tools/build_prof.py 4000000`).
At 4 M primitives: k_refit 6.3 ms, 4 x k_rs_scan 2.1 ms, k_flatten 0.8, k_karras 0.45, k_qnodes 0.3, the rest < 0.3 ms each."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from ti_raytrace_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ex = scenes.synthetic(64, 64, 4, ntri=n, spread=0.012, device_id=0); ex.build_scene(); ex.scene.ctx.sync()
for _ in range(3):
    ex.scene.ctx.lbvh_build()
print("build ms", ex.scene.ctx.stats()["ms_build"])
