"""A/B of the traversal tree option on one scene: rays/s, node visits, build time, films (must be bit-identical).
   python tools/exp/tree_ab.py [synthetic|teapot|veach|cornell] [size] [ntri]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ti_raytrace_amd import scenes, _native

which = sys.argv[1] if len(sys.argv) > 1 else "synthetic"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ntri = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
films = []
import itertools
combos = [(0, 0), (1, 0), (1, 1)] if not __import__('os').environ.get('COMBOS') else [tuple(int(c) for c in x.split(':')) for x in __import__('os').environ['COMBOS'].split(',')]
for tree, collapse in combos:
    if which == "synthetic":
        ex = scenes.synthetic(W, W, 4, ntri=ntri, device_id=0)
    else:
        ex = {"teapot": scenes.single_model, "veach": scenes.veach_bdpt, "cornell": scenes.cornell_box}[which](W, W, 4, device_id=0)
    ctx = ex.scene.ctx
    ctx.set_option("traversal_tree", tree)
    ctx.set_option("wide_collapse", collapse)
    ex.build_scene()
    build_ms = ctx.stats()["ms_build"]
    frames, steps = 32, 6
    ctx.film_clear()
    ctx.pt_rgb_render(0, frames, 1, 15, 64, 0); ctx.sync()
    ctx.stats_reset()
    t0 = time.time()
    for s in range(steps):
        ctx.pt_rgb_render(frames * (s + 1), frames, 1, 15, 64, 0)
    ctx.sync()
    dt = time.time() - t0
    st = ctx.stats(); rays = st["rays_closest"] + st["rays_shadow"]
    ctx.stats_reset()
    ctx.pt_rgb_render(frames * 20, 4, 1, 15, 64, _native.COUNT_NODES); ctx.sync()
    c = ctx.stats(); r2 = c["rays_closest"] + c["rays_shadow"]
    films.append(ctx.film_download(W, W)[0])
    print("%s tree=%d collapse=%d  build %.3f ms   %8.1f Mrays/s   %.2f node visits, %.2f prim tests per ray   wide nodes %d" %
          (which, tree, collapse, build_ms, rays / dt / 1e6, (c["box_closest"] + c["box_shadow"]) / 4.0 / r2, (c["leaf_closest"] + c["leaf_shadow"]) / r2, ctx.bvh_info()["nodes"]))
    del ex, ctx
a = films[0]
for b in films[1:]:
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print("films bit-identical:", same, "" if same else "differing pixels: %d" % int((a.view(np.uint32) != b.view(np.uint32)).any(axis=-1).sum()))
