"""Schedule of one k_trace launch on the headline scene: when its waves start, find the ray queue empty and end, and where they ran.
usage: python tools/timeline.py [launch_index ...]   (counting launches of one 32-frame step; 0 = camera rays, 1.. = bounces)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from ti_raytrace_amd import scenes, _native

import os
launches = [int(a) for a in sys.argv[1:]] or [0, 1, 3, 8]
ex = scenes.synthetic(1024, 1024, 64, ntri=100000, device_id=0, seed=1)
ctx = ex.scene.ctx
for kv in os.environ.get("TIRT_OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
ex.build_scene(); ctx.sync()
ctx.pt_rgb_render(0, 32, 1, 15, 64, 0); ctx.sync()
ctx.set_option("time_kernels", 1)
for k in launches:
    ctx.set_option("trace_timeline", k)
    ctx.stats_reset()
    ctx.pt_rgb_render(0, 32, 1, 15, 64, _native.TRAVERSE_ORDERED | _native.COUNT_NODES); ctx.sync()
    st = ctx.stats()
    tl = ctx.trace_timeline()
    if len(tl) == 0:
        print("launch %d: not recorded" % k); continue
    t0 = tl[:, 0].min()
    start = (tl[:, 0] - t0).astype(np.float64) * 1e-2          # us
    end = (tl[:, 2] - t0).astype(np.float64) * 1e-2
    exh = np.where(tl[:, 1] > 0, (tl[:, 1].astype(np.int64) - np.int64(t0)).astype(np.float64) * 1e-2, np.nan)
    span = end.max()
    life = end - start
    hw = (tl[:, 3] & np.uint64(0xffffffff)).astype(np.int64); xcc = (tl[:, 3] >> np.uint64(32)).astype(np.int64) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("launch %d: %d waves, span %.1f us; mean life %.1f us (%.3f of span); starts: p50 %.1f p90 %.1f max %.1f; queue empty: min %.1f p50 %.1f; "
          "ends: p10 %.1f p50 %.1f p90 %.1f" % (k, len(tl), span, life.mean(), life.mean() / span, np.percentile(start, 50), np.percentile(start, 90), start.max(),
                                               np.nanmin(exh), np.nanmedian(exh), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90)))
    short = life < 0.05 * span
    print("   waves with life < 5 %% of span: %d (%.1f %%); distinct CUs %d, waves per CU min/median/max %s; per XCC %s" % (
        short.sum(), 100.0 * short.mean(), len(np.unique(cuid)), np.percentile(np.bincount(cuid)[np.bincount(cuid) > 0], [0, 50, 100]).tolist(), np.bincount(xcc).tolist()))
    # occupancy over time: waves alive at 20 points of the span
    ts = np.linspace(0, span, 21)[:-1] + span / 40
    alive = [(int(((start <= t) & (end > t)).sum())) for t in ts]
    print("   waves alive over the span (20 bins):", alive)
    busy = [(int(((start <= t) & (end > t) & ~short).sum())) for t in ts]
    print("   kernel time per launch (events): %.1f us" % (1e3 * (st["ms_trace_closest"] + st["ms_trace_shadow"]) / max(st["launches_trace_closest"] + st["launches_trace_shadow"], 1)))
