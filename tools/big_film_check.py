"""One-off robustness check: films larger than one batch (8192^2 = 64 Mi pixels > batch_paths) render and
agree with a tiled render of the same frames (pixel tiles are independent, so the sums must match exactly)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ti_raytrace_amd import scenes
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ex = scenes.synthetic(W, H, 4, ntri=100000, device_id=0); ex.build_scene(); ctx = ex.scene.ctx
t = time.time(); ex.integrator.render_frames(2); ctx.sync(); dt = time.time() - t
st = ctx.stats(); full = ex.integrator.hdr.to_numpy()
print("%dx%d x2 frames: %.3f s, %.1f Mrays/s, finite %s, overflow %d" % (W, H, dt, (st["rays_closest"] + st["rays_shadow"]) / dt / 1e6, np.isfinite(full).all(), st["stack_overflow"]))
acc = np.zeros_like(full)
for rank in range(2):
    ctx.film_create(W, H, rank, 2, 4096); ctx.pt_rgb_render(0, 2, ex.integrator.seed, 15, 64, 0); acc += ctx.film_download(W, H)[0]
print("two pixel-tile halves add up to the full film bit for bit:", np.array_equal(acc, full))
