import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ti_raytrace_amd import scenes
tree = int(sys.argv[1]) if len(sys.argv) > 1 else 1
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ex = scenes.veach_bdpt(512, 512, 64, device_id=0)
ex.scene.ctx.set_option("traversal_tree", tree)
ex.build_scene(); ctx = ex.scene.ctx
for f in range(0, 64, step):
    ctx.bdpt_rgb_render(f, step, 1); ctx.sync()
    print("frames", f, f + step, "ok", flush=True)
