import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from ti_raytrace_amd import scenes
import oracle_api as oa
ex=scenes.synthetic(64,64,4,device_id=0); ex.build_scene(); ctx=ex.scene.ctx
r=np.random.RandomState(1)
for n in (64, 1024, 16384, 65536, 262144):
    o=r.uniform(-0.9,0.9,(n,3)); d=r.normal(size=(n,3)); d/=np.linalg.norm(d,axis=1,keepdims=True)
    rays=np.concatenate([o,d],1).astype(np.float32)
    for k in range(3): ctx.trace_closest(rays,64,0)
