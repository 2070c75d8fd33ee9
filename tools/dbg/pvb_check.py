"""Camera rays through the pixels' candidate lists (option primary_beams) against the ordinary bounce-0 launch: films bit for bit, list statistics, time.
python tools/dbg/pvb_check.py [scene ...]   scenes: synthetic cornell teapot gallery"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ti_raytrace_amd import scenes

W = H = 0
def make(name):
    global W, H
    W, H = (1024, 1024) if name in ("synthetic", "teapot") else (512, 512)
    if name == "synthetic": return scenes.synthetic(1024, 1024, 64, device_id=0), 64
    if name == "cornell": return scenes.cornell_box(512, 512, 64, device_id=0), 64
    if name == "teapot": return scenes.single_model(1024, 1024, 32, device_id=0), 32
    if name == "gallery": return scenes.gallery_sphere(512, 512, 32, device_id=0), 32
    raise SystemExit("unknown scene " + name)

for name in (sys.argv[1:] or ["synthetic", "cornell", "teapot", "gallery"]):
    films = {}
    for beams in (0, 1):
        ex, spp = make(name); ex.build_scene(); ctx = ex.scene.ctx
        ctx.set_option("primary_beams", beams)
        ctx.set_option("job_frames", spp)
        ctx.pt_rgb_render(0, spp, 1, 15); ctx.sync()      # warm-up (allocations, lists)
        ctx.film_clear(); ctx.sync(); ctx.stats_reset()
        t0 = time.perf_counter()
        ctx.pt_rgb_render(0, spp, 1, 15); ctx.sync()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        films[beams] = ctx.film_download(W, H)[0].copy()
        print("%-10s beams %d: %.4f s  %.1f Mrays/s  rays %d+%d" % (name, beams, dt, (st["rays_closest"] + st["rays_shadow"]) / dt / 1e6, st["rays_closest"], st["rays_shadow"]),
              ctx.primary_beam_stats() if beams else "")
    a, b = films[0].view(np.uint32), films[1].view(np.uint32)
    print("%-10s films identical: %s  (%d of %d words differ)" % (name, bool((a == b).all()), int((a != b).sum()), a.size))
