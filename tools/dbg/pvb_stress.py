"""Candidate lists on / off over many camera poses (orbit angles, distances from inside the scene to seven extents away, three film shapes) on a triangle soup,
the Cornell box (analytic sphere light), the glass teapot under an env map and a scene of long thin triangles: python tools/dbg/pvb_stress.py [poses per scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from ti_raytrace_amd import scenes
from common import tiny_scene

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(7)
bad = 0; runs = 0
def mk(name, W, H, n):
    if name == "soup": return scenes.synthetic(W, H, n, ntri=30000, device_id=0)
    if name == "cornell": return scenes.cornell_box(W, H, n, device_id=0)
    if name == "teapot": return scenes.single_model(W, H, n, device_id=0)
    if name == "slivers": return tiny_scene(3000, seed=11, W=W, H=H, spread=0.9, device_id=0)
for name in ("soup", "cornell", "teapot", "slivers"):
    for W, H in ((192, 192), (320, 64), (56, 200)):
        frames = 8
        ex = mk(name, W, H, frames); ex.build_scene(); ctx = ex.scene.ctx
        ctx.set_option("primary_beams_min_frames", 1)
        for pose in range(poses):
            ex.cam.yaw = float(rng.uniform(0, 6.28)); ex.cam.pitch = float(rng.uniform(-1.2, 1.2))
            ex.frame_camera(float(rng.choice([0.05, 0.2, 0.5, 0.8, 1.5, 3.0, 7.0])))
            films = []
            for beams in (0, 1):
                ctx.set_option("primary_beams", beams)
                ctx.film_clear(); ctx.stats_reset()
                ctx.pt_rgb_render(0, frames, 11 + pose, 15, 64, 0)
                st = ctx.stats()
                films.append((ctx.film_download(W, H)[0].view(np.uint32).copy(), st["rays_closest"], st["rays_shadow"], st["stack_overflow"]))
            runs += 1
            same = np.array_equal(films[0][0], films[1][0]) and films[0][1:] == films[1][1:]
            if not same:
                bad += 1
                print("DIFF", name, (W, H), "pose", pose, "yaw %.3f pitch %.3f scale %.3f" % (ex.cam.yaw, ex.cam.pitch, ex.cam.scale), int((films[0][0] != films[1][0]).sum()), films[0][1:], films[1][1:], ctx.primary_beam_stats())
        print(name, (W, H), "done", ctx.primary_beam_stats())
print("runs", runs, "different", bad)
