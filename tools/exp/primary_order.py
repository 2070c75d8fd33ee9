"""Experiment: does the order of the camera rays within a wave matter to k_trace?  The wavefront lists them column by column
(64 consecutive pixels of one column per wave); here the same rays in 4 x 16 and 8 x 8 pixel blocks per wave, and shuffled.
Run under rocprofv3 --kernel-trace and read the k_trace<0,false,0> durations in launch order (each order twice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as oa
from ti_raytrace_amd import scenes
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ex = scenes.synthetic(W, W, 4, device_id=0); ex.build_scene(); ctx = ex.scene.ctx
rays = oa.camera_rays(ex.cam, W, W).astype(np.float32)         # index p = i * W + j
i, j = np.divmod(np.arange(W * W), W)
def block_order(bi, bj):
    key = ((i // bi) * (W // bj) + (j // bj)) * (bi * bj) + (i % bi) * bj + (j % bj)
    return np.argsort(key, kind="stable")
orders = {"column strips 1x64": np.arange(W * W), "4x16": block_order(4, 16), "8x8": block_order(8, 8), "shuffled": np.random.RandomState(1).permutation(W * W)}
ref = None
for name, o in orders.items():
    for rep in range(2):
        got, gp, _ = ctx.trace_closest(rays[o], 64, 0)
    back = np.empty(W * W, np.int32); back[o] = gp
    if ref is None: ref = back
    print(name, "same hits:", bool(np.array_equal(back, ref)))
