"""Scene container (mirror of the host half of the reference's ``Scene.py``).

Host: OBJ/MTL ingest (``add_obj`` :59-141), analytic shapes (``add_shape`` :188-205), env
(``add_env`` :183-185), struct packing (``setup_data_cpu`` :223-296), uploads + LBVH build
(``setup_data_gpu`` :299-310), ``total_area`` (:747-750), ``process_normal`` (:754-798).
Device: everything the reference declares as ``@ti.func`` on this class (``closet_hit``,
``closet_hit_shadow``, ``intersect_*``, ``sample_li`` ...) is HIP code in ``csrc/`` reached
through the C-ABI (``include/tirt.h``); there is no CPU fallback.

Ingest is vectorised numpy instead of the reference's per-vertex Python loops, but the
numbers it produces (primitive order, f32 rounding points, face normals, AABB) follow the
reference; ``tests/test_oracle_golden.py`` pins that against the reference's nodelist.txt.
"""
import os

import numpy as np

from . import SceneData as SCD
from . import LBvh
from . import Texture as TX
from . import ObjLoader
from . import _native

MAX_STACK_SIZE = 32          # Scene.py:19 (process_normal's point-query stack)
INF_VALUE = 1000000.0


class DeviceField:
    """Stand-in for a Taichi field living on the device: knows its owner and how to read
    itself back (``to_numpy()``), which is all the reference's host code does with fields."""

    def __init__(self, name, owner, reader):
        self.name = name
        self._owner = owner
        self._reader = reader

    @property
    def ctx(self):
        return self._owner.ctx

    def to_numpy(self):
        return self._reader()


class Scene:
    def __init__(self, device_id=None):
        self.maxboundarynp = np.full((1, 3), -INF_VALUE, dtype=np.float32)
        self.minboundarynp = np.full((1, 3), INF_VALUE, dtype=np.float32)

        self.light_cpu = []
        self.material_cpu = []
        self.shape_cpu = []
        self._vertex_chunks = []        # float64 [k, 9] blocks in primitive order
        self._prim_mat = []             # (first_prim, count, material index) per block
        self._shape_prims = []          # (prim index, shape index, material index)

        self.material_count = 0
        self.vertex_count = 0
        self.primitive_count = 0
        self.shape_count = 0
        self.light_count = 0

        self.env = TX.Texture()
        self.env_power = 0.0
        self.bvh = None

        self._device_id = device_id
        self._ctx = None
        self._light_area = np.zeros(1, np.float32)

        self.vertex = DeviceField("vertex", self, lambda: self.ctx.vertex_download(self.vertex_count))
        self.primitive = DeviceField("primitive", self, lambda: self.primitive_np.copy())
        self.shape = DeviceField("shape", self, lambda: self.shape_np.copy())
        self.material = DeviceField("material", self, lambda: self.material_np.copy())
        self.light = DeviceField("light", self, lambda: self.light_np.copy())
        self.light_area = DeviceField("light_area", self, lambda: self._light_area.copy())

    # -- device context ---------------------------------------------------------------------
    @property
    def ctx(self):
        if self._ctx is None:
            dev = self._device_id
            if dev is None:
                dev = int(os.environ.get("LOCAL_RANK", "0"))
            self._ctx = _native.Context(dev)
        return self._ctx

    # -- ingest -------------------------------------------------------------------------------
    def add_obj(self, filename):
        """Scene.py:59-141.  One material per MTL entry in file order; its faces become
        consecutive triangles (9-float vertex rows: pos, normal, uv+0)."""
        scene = ObjLoader.Wavefront(filename)
        for name in scene.materials:
            src = scene.materials[name]
            material = SCD.Material()
            if (src.emissive[0] > 1.0) and (src.emissive[1] > 1.0) and (src.emissive[2] > 1.0):
                material.type = SCD.MAT_LIGHT
                material.setColor(src.emissive)
            elif src.transparency > 0.99:
                material.type = SCD.MAT_DISNEY
                material.setMetal(0.0)
                material.setRough(0.5)
                material.setColor(src.diffuse)
            else:
                material.type = SCD.MAT_GLASS
                material.setIor(src.optical_density)
                material.setExtinciton(src.shininess)
                material.setColor(src.diffuse)
            material.alebdoTex = -1
            self.material_cpu.append(material)

            flat = src.vertices
            stride = src.vertex_size
            if stride and flat.size:
                rows = flat.reshape(-1, stride)
                rows = rows[: (rows.shape[0] // 3) * 3]
                block = np.zeros((rows.shape[0], SCD.VER_VEC_SIZE), dtype=np.float64)
                fmt = src.vertex_format
                block[:, 0:3] = rows[:, stride - 3:stride]
                if fmt.startswith("T2F"):
                    block[:, 6:8] = rows[:, 0:2]
                if "N3F" in fmt:
                    off = 2 if fmt.startswith("T2F") else 0
                    block[:, 3:6] = rows[:, off:off + 3]
                self._add_block(block, self.material_count, material.type == SCD.MAT_LIGHT)
            self.material_count += 1

    def add_mesh(self, positions, material, normals=None):
        """Extension (no reference equivalent): append a triangle soup ``positions[k,3,3]``
        with one material -- what ``add_obj`` would produce for a one-material OBJ without
        reading a file.  Used for the synthetic headline scene (BASELINE config 3)."""
        positions = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        block = np.zeros((positions.shape[0], SCD.VER_VEC_SIZE), dtype=np.float64)
        block[:, 0:3] = positions
        if normals is not None:
            block[:, 3:6] = np.asarray(normals, dtype=np.float64).reshape(-1, 3)
        self.material_cpu.append(material)
        self._add_block(block, self.material_count, material.type == SCD.MAT_LIGHT)
        self.material_count += 1

    def _add_block(self, block, mat_index, is_light):
        ntri = block.shape[0] // 3
        if ntri == 0:
            return
        pos = block[:, 0:3]
        self.maxboundarynp[0, :] = np.maximum(self.maxboundarynp[0, :].astype(np.float64), pos.max(axis=0))
        self.minboundarynp[0, :] = np.minimum(self.minboundarynp[0, :].astype(np.float64), pos.min(axis=0))
        self._vertex_chunks.append(block)
        self._prim_mat.append((self.primitive_count, ntri, mat_index, self.vertex_count))
        if is_light:
            self.light_cpu.extend(range(self.primitive_count, self.primitive_count + ntri))
            self.light_count += ntri
        self.vertex_count += 3 * ntri
        self.primitive_count += ntri

    def add_env(self, filename, env_power):
        # The reference ignores `filename` and always loads image/env.png (Scene.py:183-185,
        # quirk B16); here the argument is honoured.
        self.env.load_image(filename)
        self.env_power = env_power

    def add_shape(self, shape, mat):
        """Scene.py:188-205.  Does not extend the scene AABB (quirk B8)."""
        if mat.type == SCD.MAT_LIGHT:
            self.light_cpu.append(self.primitive_count)
            self.light_count += 1
        self._shape_prims.append((self.primitive_count, self.shape_count, self.material_count))
        self.primitive_count += 1
        self.shape_cpu.append(shape)
        self.shape_count += 1
        self.material_cpu.append(mat)
        self.material_count += 1

    # -- packing --------------------------------------------------------------------------------
    def cal_normal(self, verts):
        """Scene.py:169-179: triangles whose first vertex has a zero normal get the face
        normal normalize((v1-v0) x (v2-v0)) on all three vertices (double precision)."""
        tri = verts.reshape(-1, 3, SCD.VER_VEC_SIZE)
        n0 = tri[:, 0, 3:6]
        need = np.sqrt(n0[:, 0] * n0[:, 0] + n0[:, 1] * n0[:, 1] + n0[:, 2] * n0[:, 2]) == 0.0
        if not need.any():
            return
        a = tri[need, 1, 0:3] - tri[need, 0, 0:3]
        b = tri[need, 2, 0:3] - tri[need, 0, 0:3]
        n = np.stack([a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1],
                      a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2],
                      a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]], axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / np.sqrt(n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2])
        n = n * inv[:, None]
        for k in range(3):
            tri[need, k, 3:6] = n

    def setup_data_cpu(self):
        """Scene.py:223-296: host structs -> packed numpy rows; declares the LBVH."""
        self.material_np = np.zeros((self.material_count, SCD.MAT_VEC_SIZE), dtype=np.float32)
        for i, m in enumerate(self.material_cpu):
            m.fillStruct(self.material_np, i)

        if self._vertex_chunks:
            verts = np.concatenate(self._vertex_chunks, axis=0)
        else:
            verts = np.zeros((0, SCD.VER_VEC_SIZE), dtype=np.float64)
        self.cal_normal(verts)
        self.vertex_np = verts.astype(np.float32)
        self.smooth_normal_np = np.zeros((self.vertex_count, 3), dtype=np.float32)

        self.primitive_np = np.zeros((self.primitive_count, SCD.PRI_VEC_SIZE), dtype=np.int32)
        self.vertex_index_np = np.zeros(self.vertex_count, dtype=np.int32)
        for first, ntri, mat, vfirst in self._prim_mat:
            ids = np.arange(ntri, dtype=np.int32)
            self.primitive_np[first:first + ntri, 0] = SCD.PRIMITIVE_TRI
            self.primitive_np[first:first + ntri, 1] = vfirst + 3 * ids
            self.primitive_np[first:first + ntri, 2] = mat
            self.vertex_index_np[vfirst:vfirst + 3 * ntri] = first + np.repeat(ids, 3)
        for prim, sha, mat in self._shape_prims:
            self.primitive_np[prim] = (SCD.PRIMITIVE_SHAPE, sha, mat)

        if self.light_count > 0:
            self.light_np = np.asarray(self.light_cpu, dtype=np.int32)
        else:
            self.light_np = np.zeros(1, dtype=np.int32)          # Scene.py:259-261

        if self.shape_count > 0:
            self.shape_np = np.zeros((self.shape_count, SCD.SHA_VEC_SIZE), dtype=np.float32)
            for i, s in enumerate(self.shape_cpu):
                s.fillStruct(self.shape_np, i)
        else:
            self.shape_np = np.zeros((1, SCD.SHA_VEC_SIZE), dtype=np.float32)

        self.bvh = LBvh.Bvh(self.primitive_count, self.minboundarynp, self.maxboundarynp)
        self.bvh.setup_data_cpu()

        if self.env_power == 0.0:
            self.env.load_black()                                    # Scene.py:295-296

    def setup_data_gpu(self):
        """Scene.py:299-310: uploads, then the device LBVH build."""
        ctx = self.ctx
        ctx.scene_upload(self.vertex_np, self.primitive_np, self.material_np, self.shape_np,
                         self.light_np, self.light_count, self.minboundarynp, self.maxboundarynp)
        self.env.setup_data_gpu(ctx, self.env_power)
        self.bvh.setup_data_gpu(self.vertex, self.shape, self.primitive)

    # -- kernels ------------------------------------------------------------------------------------
    def total_area(self):
        """Scene.py:747-750 (accumulates, like the reference's ``+=``)."""
        self._light_area[0] += np.float32(self.ctx.total_area())

    def process_normal(self):
        """Scene.py:754-798: angle x area weighted smooth normals via a BVH point query."""
        self.ctx.process_normal(self.vertex_index_np)
        self.normals_processed = True
