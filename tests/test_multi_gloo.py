"""N > 1 path on CPU: 2 processes, gloo.  Each rank renders its pixel tiles (here with the
oracle standing in for the device, same tiling rule and same film layout), the films are
sum-reduced through ti_raytrace_amd.distributed onto rank 0 and must equal the single-process
film bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import oracle_api as oa
from common import host_only
from ti_raytrace_amd import scenes, distributed as tdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
W = H = 40
ex = host_only(scenes.cornell_box(W, H, 4))
o = oa.OracleScene(ex.scene, ex.cam); o.lbvh_build()
part, st = o.render(W, H, 0, 3, tile_rank=rank, tile_count=world, tile_size=64, nthreads=2)
assert st["paths"] == 3 * tdist.local_pixel_count(W, H, rank, world, 64)
owned = (np.arange(W * H) // 64) % world == rank
assert (part.reshape(-1, 3)[~owned] == 0).all()
film = torch.from_numpy(part.copy())
tdist.reduce_film_tensor(film, dst=0)
if rank == 0:
    full, _ = o.render(W, H, 0, 3, nthreads=2)
    assert np.array_equal(film.numpy(), full), "reduced film differs from the single-process film"
    print("GLOO_OK")
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_tile_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GLOO_OK" in out.stdout


def test_tile_bookkeeping():
    from ti_raytrace_amd import distributed as tdist
    W, H = 37, 29
    for world in (1, 2, 3, 8):
        for ts in (1, 64, 100, 4096):
            counts = [tdist.local_pixel_count(W, H, r, world, ts) for r in range(world)]
            assert sum(counts) == W * H
            owners = tdist.tile_owner(np.arange(W * H), ts, world)
            assert [int((owners == r).sum()) for r in range(world)] == counts
