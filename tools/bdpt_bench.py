"""BASELINE config 5 (veach_bdpt.py, BDPT_RGB 512x512) timed after a warm-up batch: python tools/bdpt_bench.py [spp] [size]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ti_raytrace_amd import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
opts = sys.argv[3:]
ex = scenes.veach_bdpt(size, size, spp, device_id=0)
ex.build_scene(); ctx = ex.scene.ctx
for kv in opts:
    k, v = kv.split("="); ctx.set_option(k, float(v))
ctx.bdpt_rgb_render(0, spp, 1); ctx.sync()                      # warm-up with the job's own size (allocations), then a fresh film
ctx.film_clear(); ctx.sync(); ctx.stats_reset()
t0 = time.perf_counter()
ctx.bdpt_rgb_render(0, spp, 1); ctx.sync()
dt = time.perf_counter() - t0
st = ctx.stats()
rays = st["rays_closest"] + st["rays_shadow"]
print(json.dumps({"config": "veach_bdpt %dx%d x%d spp" % (size, size, spp), "seconds": round(dt, 4), "Mrays_per_s": round(rays / dt / 1e6, 1),
                  "rays_closest": st["rays_closest"], "rays_shadow": st["rays_shadow"], "opts": opts}))
