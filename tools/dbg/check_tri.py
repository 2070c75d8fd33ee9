import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ti_raytrace_amd import scenes, _native
ex = scenes.cornell_box(48, 48, 4, device_id=0); ex.build_scene(); ctx = ex.scene.ctx
n = ex.scene.primitive_count; N = 2 * n - 1
lib = _native.lib(); lib.tirt_exp_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
def dl(which, shape, dt):
    a = np.zeros(shape, dt); rc = lib.tirt_exp_download(ctx.handle, which, a.ctypes.data_as(C.c_void_p), a.nbytes); assert rc == 0; return a
tri = dl(0, (n, 12), np.float32); slot = dl(1, n, np.int32); lc = dl(2, n, np.int32); cparent = dl(4, N, np.int32)
print("slot", slot)
print("perm ok", np.array_equal(np.sort(slot), np.arange(n)))
ids = tri[:, 11].view(np.int32); leaf = tri[:, 3].view(np.int32)
print("ids by slot", ids)
print("id ok", np.array_equal(ids[slot], np.arange(n)), "leaf ok", np.array_equal(leaf[slot], lc))
print("leaf", leaf, "N", N)
cn = dl(3, (ctx.bvh_info()["nodes"], 16), np.uint32)
codes = cn[:, 12:].view(np.int32)
lf = codes[(codes < 0) & (codes != -2147483647)]
print("leaf slots in cnode", np.sort((~lf) & 0x3fffffff))
