"""BDPT_RGB device kernels against the CPU oracle over seeds, film shapes and batch splits (GPU box):
    python tools/stress_bdpt.py [n_seeds]
Per case: equal ray counts, equal non-finite masks, rel-L2 of the films (float-atomic splats: ~1e-7)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as oa
from ti_raytrace_amd import scenes, BDPT_RGB

def rel_l2(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
worst = 0.0; bad = 0; cases = 0
for scene in ("veach", "cornell"):
    for (W, H, frames) in ((40, 40, 5), (56, 32, 4), (24, 72, 6)):
        for seed in range(1, n_seeds + 1):
            if scene == "veach":
                ex = scenes.veach_bdpt(W, H, frames, device_id=0, seed=seed)
            else:
                ex = scenes.cornell_box(W, H, frames, device_id=0, seed=seed)
                ex.integrator = BDPT_RGB.BDPT(W, H, ex.cam, ex.scene, 64); ex.integrator.seed = seed
            ex.build_scene()
            o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
            if scene == "veach": o.process_normal(ex.scene.vertex_index_np)
            ctx = ex.scene.ctx
            ctx.set_option("bdpt_state_fill", 2)
            ctx.set_option("bdpt_batch_items", W * H * (1 + seed % 3))          # 1..3 frames per batch: chunk and batch seams move
            ctx.stats_reset()
            ctx.bdpt_rgb_render(0, frames, seed)
            st = ctx.stats(); got = ctx.film_download(W, H)[0]
            want, ost, _ = o.bdpt_render(ex.cam, W, H, 0, frames, seed=seed)
            fm = np.isfinite(want).all(axis=2); gm = np.isfinite(got).all(axis=2)
            r = rel_l2(got[fm & gm], want[fm & gm])
            ok = (st["rays_closest"] == ost["rays_closest"] and st["rays_shadow"] == ost["rays_shadow"] and (fm == gm).all() and r <= 1e-5)
            cases += 1; bad += 0 if ok else 1; worst = max(worst, r)
            print("%-8s %3dx%-3d x%d seed %d: rays %d/%d  rel-L2 %.2e  %s" % (scene, W, H, frames, seed, st["rays_closest"], st["rays_shadow"], r, "ok" if ok else "MISMATCH"), flush=True)
            ctx.close()
print("cases %d, mismatches %d, worst rel-L2 %.2e" % (cases, bad, worst))
