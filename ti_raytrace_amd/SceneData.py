"""Packed-struct vocabulary shared by host and device.

Mirrors the constants and host classes of the reference's ``SceneData.py`` (:33-214) --
same names (typos included: ``SHPAE_*``, ``alebdoTex``, ``setExtinciton``) and the same
``fillStruct(np_data, index)`` packing, because example scripts poke these objects
directly (``scene.material_cpu[0].setIor(1.3)``, ``shape.setRadius(5.0)``).

Row layouts (f32 unless noted), reference SceneData.py:7-30:
  material  [10]: type, albedoTex, r, g, b, param0..4   (disney: metallic, roughness;
                                                         glass: ior, extinction)
  shape     [10]: type, px, py, pz, param0..5           (sphere: radius)
  vertex    [ 9]: pos3, normal3, uv3
  primitive [ 3] i32: type (1 tri / 2 shape), first-vertex | shape index, material index
  bvh_node  [11]: flags, left, right, parent, prim, min3, max3      (ints stored as f32)
  compact   [ 9]: flags, prim | right-child offset, min3, max3, unused
"""
import numpy as np

MAT_VEC_SIZE = 10
VER_VEC_SIZE = 9
PRI_VEC_SIZE = 3
SHA_VEC_SIZE = 10
NOD_VEC_SIZE = 11
CPNOD_VEC_SIZE = 9

SHPAE_NONE, SHPAE_SPHERE, SHPAE_QUAD, SHPAE_SPOT, SHPAE_LASER = 0, 1, 2, 3, 4
PRIMITIVE_NONE, PRIMITIVE_TRI, PRIMITIVE_SHAPE = 0, 1, 2

MAT_DISNEY = 0.0
MAT_GLASS = 1.0
MAT_LIGHT = 2.0
MAT_SPECTRAL = 10.0

IS_LEAF = 1


class Material:
    """type / alebdoTex / color[3+] / param[5]  ->  material row (SceneData.py:57-86)."""

    def __init__(self):
        self.type = 0
        self.alebdoTex = 0
        self.color = [0.0, 0.0, 0.0]
        self.param = [0.0] * 5

    def setColor(self, color):
        self.color = color

    def setMetal(self, metal):
        self.param[0] = metal

    def setRough(self, rough):
        self.param[1] = rough

    def setIor(self, ior):
        self.param[0] = ior

    def setExtinciton(self, extinction):
        self.param[1] = extinction

    def fillStruct(self, np_data, index):
        row = np_data[index]
        row[0] = float(self.type)
        row[1] = float(self.alebdoTex)
        row[2:5] = [self.color[0], self.color[1], self.color[2]]
        row[5:MAT_VEC_SIZE] = self.param[:MAT_VEC_SIZE - 5]


class Shape:
    """type / pos[3] / param[6]  ->  shape row (SceneData.py:88-129)."""

    def __init__(self):
        self.type = 0
        self.pos = [0.0, 0.0, 0.0]
        self.param = [0.0] * 6

    def setRadius(self, radius):
        self.param[0] = radius

    def getRadius(self):
        return self.param[0]

    def setXita(self, xita1, xita2):
        self.param[0], self.param[1] = xita1, xita2

    def setScale(self, scale):
        self.param[2] = scale

    def setV1(self, V1):
        self.param[0:3] = [V1[0], V1[1], V1[2]]

    def setV2(self, V2):
        self.param[3:6] = [V2[0], V2[1], V2[2]]

    def setNormal(self, normal):
        self.param[3:6] = [normal[0], normal[1], normal[2]]

    def fillStruct(self, np_data, index):
        row = np_data[index]
        row[0] = float(self.type)
        row[1:4] = self.pos
        row[4:SHA_VEC_SIZE] = self.param


class Vertex:
    """pos / normal / tex  ->  vertex row (SceneData.py:132-163)."""

    def __init__(self):
        self.pos = [0.0, 0.0, 0.0]
        self.normal = [0.0, 0.0, 0.0]
        self.tex = [0.0, 0.0, 0.0]

    def setPos(self, buf, offset):
        self.pos = [buf[offset], buf[offset + 1], buf[offset + 2]]

    def setNormal(self, buf, offset):
        self.normal = [buf[offset], buf[offset + 1], buf[offset + 2]]

    def setTex(self, buf, offset):
        self.tex = [buf[offset], buf[offset + 1], 0.0]

    def setTex3(self, buf, offset):
        self.tex = [buf[offset], buf[offset + 1], buf[offset + 2]]

    def fillStruct(self, np_data, index):
        np_data[index, 0:3] = self.pos
        np_data[index, 3:6] = self.normal
        np_data[index, 6:9] = self.tex


class Primitive:
    """type / vertex_shape_index / mat_index  ->  primitive row (SceneData.py:165-174)."""

    def __init__(self):
        self.type = 0
        self.vertex_shape_index = 0
        self.mat_index = 0

    def fillStruct(self, np_data, index):
        np_data[index] = (self.type, self.vertex_shape_index, self.mat_index)


class Bounds:
    """Axis-aligned box accumulator (SceneData.py:184-203)."""

    def __init__(self):
        self.min_v3 = [np.inf] * 3
        self.max_v3 = [-np.inf] * 3

    def Merge(self, v):
        for k in range(3):
            self.min_v3[k] = min(self.min_v3[k], v[k])
            self.max_v3[k] = max(self.max_v3[k], v[k])

    def MergeBox(self, b):
        self.Merge(b.min_v3)
        self.Merge(b.max_v3)

    def GetSurfaceArea(self):
        e = [self.max_v3[k] - self.min_v3[k] for k in range(3)]
        return 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0])


class BVHNode:
    """Host view of one bvh_node row (SceneData.py:205-214)."""

    def __init__(self):
        self.is_leaf = 0
        self.axis = 0
        self.left_node = 0
        self.right_node = 0
        self.parent_node = 0
        self.prim_index = 0
        self.min_v3 = [np.inf] * 3
        self.max_v3 = [-np.inf] * 3

    @classmethod
    def from_row(cls, row):
        """Decode one 11-float bvh_node row (UtilsFunc.py:216-289 accessors)."""
        nd = cls()
        nd.is_leaf = int(row[0]) & 1
        nd.axis = (int(row[0]) & 6) >> 1
        nd.left_node, nd.right_node = int(row[1]), int(row[2])
        nd.parent_node, nd.prim_index = int(row[3]), int(row[4])
        nd.min_v3 = [float(x) for x in row[5:8]]
        nd.max_v3 = [float(x) for x in row[8:11]]
        return nd
