"""Linear BVH (mirror of the host half of the reference's ``accel/LBvh.py``).

``Bvh.setup_data_gpu`` replaces the reference's 1 110-launch build (Morton kernel, 30
one-bit radix passes with Blelloch scans, Karras kernel, ~depth refit launches with a host
sync each, and a recursive *Python* flatten; accel/LBvh.py:192-226) with one call into
``tirt_lbvh_build``: a handful of HIP kernels, no host round trips.  The products are the
reference's own: ``morton_code_s`` (sorted pairs), ``bvh_node`` (11 f32) and ``compact_node``
(9 f32, DFS order) -- bit-identical to the CPU oracle (tests/test_gpu_lbvh.py).
"""
import numpy as np

from . import SceneData as SCD

HIT_TRI = 0.0
HIT_SHA = 1.0


class _Field:
    def __init__(self, bvh, which):
        self._bvh, self._which = bvh, which

    def to_numpy(self):
        b = self._bvh
        m, n, c = b.ctx.lbvh_download(b.primitive_count, self._which == 0, self._which == 1, self._which == 2)
        return (m, n, c)[self._which]


class Bvh:
    def __init__(self, primitive_count, min_boundary, max_boundary):
        self.primitive_count = primitive_count
        self.minboundarynp = min_boundary
        self.maxboundarynp = max_boundary
        self.leaf_node_count = 0
        self.ctx = None
        self.morton_code_s = _Field(self, 0)
        self.bvh_node = _Field(self, 1)
        self.compact_node = _Field(self, 2)

    def get_pot_num(self, num):          # accel/LBvh.py:39-43
        m = 1
        while m < num:
            m <<= 1
        return m >> 1

    def get_pot_bit(self, num):          # accel/LBvh.py:46-52
        m, cnt = 1, 0
        while m < num:
            m <<= 1
            cnt += 1
        return cnt

    def setup_data_cpu(self):            # accel/LBvh.py:177-188
        self.node_count = self.primitive_count * 2 - 1
        self.primitive_pot = self.get_pot_num(self.primitive_count) << 1
        self.primitive_bit = self.get_pot_bit(self.primitive_pot)

    def setup_data_gpu(self, vertex, shape, primitive):
        """accel/LBvh.py:192-226.  ``vertex`` / ``shape`` / ``primitive`` are the scene's
        device fields (already uploaded); the build runs on their context."""
        self.ctx = vertex.ctx
        self.ctx.lbvh_build()

    def dump_nodelist(self, path="nodelist.txt"):
        """What accel/LBvh.py:164-172 writes as a side effect of every build (explicit here).
        Uses the reference's stale printer layout (:127-136) so files compare line by line."""
        with open(path, "w") as fo:
            fo.write(format_nodelist(self.compact_node.to_numpy()))


def format_nodelist(compact):
    lines = []
    for i in range(compact.shape[0]):
        r = compact[i]
        lines.append("node:%d pri:%d offset:%d min:%.2f %.2f %.2f max:%.2f %.2f %.2f"
                     % (i, int(r[1]), int(r[2]), r[3], r[4], r[5], r[6], r[7], r[8]))
    return "\n".join(lines) + "\n"
