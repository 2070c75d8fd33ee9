"""Latency probe: closest-hit kernel on a handful of rays (1 block) vs the same rays spread one per wave."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ti_raytrace_amd import scenes, _native
ex = scenes.synthetic(64, 64, 4, device_id=0); ex.build_scene(); ctx = ex.scene.ctx
r = np.random.RandomState(1)
def rays(n):
    o = r.uniform(-1.2, 1.2, size=(n, 3)); d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)
for n in (1, 8, 64, 256, 4096, 65536, 1 << 20):
    R = rays(n)
    out, prim, cnt = ctx.trace_closest(R, 64, _native.COUNT_NODES)
    ctx.trace_closest(R, 64, 0)
    t = time.perf_counter()
    for _ in range(5): ctx.trace_closest(R, 64, 0)
    dt = (time.perf_counter() - t) / 5
    print("n=%8d  wall/call %.3f ms  max node visits (ordered) %d  mean %.1f" % (n, dt * 1e3, cnt[:, 0].max() // 2, cnt[:, 0].mean() / 2))
