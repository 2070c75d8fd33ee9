"""Runs BASELINE.json configs 1-3 and 5 on one GPU and prints one JSON line per config
(Mrays/s, ray counts, LBVH build ms); writes tone-mapped PNGs to gpurun_out/."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ti_raytrace_amd import scenes
from ti_raytrace_amd import UtilsFunc as UF
from ti_raytrace_amd.Example import write_png

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(OUT, exist_ok=True)


def run(name, ex, spp, batch, warm=True):
    t0 = time.perf_counter(); ex.build_scene(); ctx = ex.scene.ctx; ctx.sync(); t_build = time.perf_counter() - t0
    # warm-up: one full batch (allocates the lanes' buffers), then start over from frame 0 on a cleared film
    if warm:
        ex.integrator.render_frames(batch); ctx.sync(); ctx.film_clear(); ctx.sync()
    ctx.stats_reset(); t0 = time.perf_counter()
    while ex.cam.frame < spp:
        k = min(batch, spp - ex.cam.frame); ex.integrator.render_frames(k); ex.cam.update_frame(k)
    ctx.sync(); dt = time.perf_counter() - t0
    st = ctx.stats()
    UF.tone_map(0.5, ex.integrator.hdr, ex.integrator.rgb_film)
    rgb = ex.integrator.rgb_film.to_numpy()
    write_png(rgb, os.path.join(OUT, name + ".png"))
    hdr = ex.integrator.hdr.to_numpy()
    rays = st["rays_closest"] + st["rays_shadow"]
    print(json.dumps({"config": name, "spp": spp, "prims": ex.scene.primitive_count, "Mrays_per_s": round(rays / dt / 1e6, 1),
                      "seconds": round(dt, 3), "rays_closest": st["rays_closest"], "rays_shadow": st["rays_shadow"],
                      "rays_per_path": round(rays / max(st["paths"], 1), 3), "lbvh_build_ms": round(st["ms_build"], 3),
                      "setup_wall_s": round(t_build, 3), "finite": bool(np.isfinite(hdr).all()),
                      "nan_pixels": int(np.isnan(hdr).any(axis=2).sum()), "inf_pixels": int(np.isinf(hdr).any(axis=2).sum()),
                      "mean_srgb": [round(float(x), 4) for x in rgb.reshape(-1, 3).mean(0)], "stack_overflow": st["stack_overflow"]}))


if __name__ == "__main__":
    which = sys.argv[1:] or ["1", "2", "3", "5"]
    if "1" in which:
        run("cfg1_cornell_512_512spp", scenes.cornell_box(512, 512, 512, device_id=0), 512, 64)
    if "2" in which:
        run("cfg2_teapot_1024_64spp", scenes.single_model(1024, 1024, 64, device_id=0), 64, 16)
    if "5" in which:
        run("cfg5_veach_bdpt_512_64spp", scenes.veach_bdpt(512, 512, 64, device_id=0), 64, 64)   # (film_clear also resets BDPT's per-pixel state)
        run("cfg5_veach_pt_512_64spp", scenes.veach_bdpt(512, 512, 64, device_id=0, integrator="pt"), 64, 32)
    if "3" in which:
        run("cfg3_synth100k_1024_256spp", scenes.synthetic(1024, 1024, 256, device_id=0), 256, 32)
