import sys, numpy as np
sys.path.insert(0, ".")
from ti_raytrace_amd import scenes, _native
ex = scenes.synthetic(1024, 1024, 64, ntri=100000, device_id=0, seed=1)
ctx = ex.scene.ctx
ctx.set_option("overlap_lanes", 1)
ex.build_scene(); ctx.sync()
ctx.pt_rgb_render(0, 32, 1, 15, 64, 0); ctx.sync()
ctx.stats_reset()
ctx.pt_rgb_render(0, 32, 1, 15, 64, _native.TRAVERSE_ORDERED | _native.COUNT_NODES); ctx.sync()
st = ctx.stats()
it = st["diag_refills"]; act = st["diag_it_outer"]; d1 = st["box_shadow"]; d2 = st["leaf_shadow"]; d4 = st["leaf_closest"]
print("drain outer iterations (all waves, 16 launches): %d = %.1f per wave-launch; busy lanes per iteration %.2f; with >= 1 stack entry %.2f, >= 2: %.2f, >= 4: %.2f"
      % (it, it / (st["diag_waves"] or 1), act / it, d1 / it, d2 / it, d4 / it))
print("wave ticks %d drain ticks %d (%.3f)" % (st["diag_wave_ticks"], st["diag_drain_ticks"], st["diag_drain_ticks"] / st["diag_wave_ticks"]))
