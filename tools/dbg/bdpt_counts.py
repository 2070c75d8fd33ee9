import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as oa
from ti_raytrace_amd import scenes
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ex = scenes.veach_bdpt(W, H, 8, device_id=0); ex.build_scene(); ctx = ex.scene.ctx
o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build(); o.process_normal(ex.scene.vertex_index_np)
state = None; hdr = None
prev_g = (0, 0); prev_o = (0, 0)
ctx.stats_reset()
tot_o = [0, 0]
for f in range(6):
    ctx.bdpt_rgb_render(f, 1, 1); st = ctx.stats()
    if f == 5: os.environ["ORC_BDPT_DUMP"] = "/tmp/orc_bdpt_dump.txt"
    hdr, ost, state = o.bdpt_render(ex.cam, W, H, f, 1, seed=1, hdr=hdr, state=state)
    tot_o[0] += ost["rays_closest"]; tot_o[1] += ost["rays_shadow"]
    g = (st["rays_closest"], st["rays_shadow"])
    print("frame", f, "device +%d +%d" % (g[0] - prev_g[0], g[1] - prev_g[1]), "oracle +%d +%d" % (ost["rays_closest"], ost["rays_shadow"]))
    prev_g = g
got = ctx.film_download(W, H)[0]
m = np.isfinite(got).all(axis=2) & np.isfinite(hdr).all(axis=2)
print("nan pixels device", (~np.isfinite(got).all(axis=2)).sum(), "oracle", (~np.isfinite(hdr).all(axis=2)).sum(), "rel", np.linalg.norm(got[m] - hdr[m]) / np.linalg.norm(hdr[m]))
# per-pixel connection rays of frame 5: device (icount of the last single-frame batch) vs oracle (ORC_BDPT_DUMP)
import ctypes as C
from ti_raytrace_amd import _native
lib = _native.lib(); lib.tirt_exp_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
BD_PAIRS = int(os.environ.get("BD_PAIRS", "36"))
P = W * H
q = np.zeros(P * (BD_PAIRS + 2), np.int32)
assert lib.tirt_exp_download(ctx.handle, 5, q.ctypes.data_as(C.c_void_p), q.nbytes) == 0
icount = q[P * BD_PAIRS + P:]
od = np.loadtxt("/tmp/orc_bdpt_dump.txt", dtype=np.int64)
oc = od[:, 1]
# device items are indexed by local pixel k -> pixel p through the tile map (one tile set: identity for tile_count 1?)
bad = np.flatnonzero(icount != oc)
print("pixels with different connection-ray counts:", bad[:20], "device", icount[bad[:20]], "oracle", oc[bad[:20]], "depths", od[bad[:20], 2:])
nz = np.flatnonzero(q)
print("qidx words", q.size, "nonzero", nz.size, "first nonzero at", nz[:5], "P*BD_PAIRS", P * BD_PAIRS, "sum icount region", q[P * BD_PAIRS + P:].sum(), "sum ibase region", q[P * BD_PAIRS:P * BD_PAIRS + P].sum())
for cand in (36, 42, 49):
    ic = q[P * cand + P: P * cand + 2 * P] if P * cand + 2 * P <= q.size else None
    if ic is not None: print("BD_PAIRS", cand, "icount sum", ic.sum(), "oracle", oc.sum())
