"""Wavefront OBJ/MTL reader with PyWavefront 1.3.3 grouping semantics.

The reference ingests meshes through the third-party ``pywavefront`` package
(``Scene.py:66-127``; pinned 1.3.3 in requirements.txt l.179), which is not available here
(SURVEY.md fact 0.2).  What ``Scene.add_obj`` relies on is restated here:

* materials appear in MTL-file order (dict insertion order); a face seen before any
  ``usemtl`` creates a default material on the spot;
* every material owns ONE interleaved float list; all faces that use the material are
  appended to it in file order, across ``o``/``g``/``usemtl`` blocks;
* the vertex format is chosen from what the face tokens carry: ``T2F_N3F_V3F``,
  ``T2F_V3F``, ``N3F_V3F`` or ``V3F``;
* polygons are fan-triangulated as (v0, v[k-1], v[k]);
* material defaults: diffuse .8, emissive 0, transparency (``d``) 1.0, optical_density
  (``Ni``) 1.0, shininess (``Ns``) 0.0, texture None.

The reference's committed ``nodelist.txt`` pins the material order + grouping for
``cornell_box.obj`` (tests/test_oracle_golden.py); the fan order is unpinned.

``Wavefront(path)`` parses with the native reader of libtirt.so (``tirt_obj_load``,
csrc/tirt_obj.hip: host C++, 3-7x faster: Teapot.obj 0.13 s -> 0.03 s); ``Wavefront(path, native=False)`` runs
the pure-Python parser below, which tests/test_host.py uses as the checker of the native one
(same materials, same doubles).
"""
import os

import numpy as np


class ObjMaterial:
    def __init__(self, name, is_default=False):
        self.name = name
        self.is_default = is_default
        self.diffuse = [0.8, 0.8, 0.8, 1.0]
        self.ambient = [0.2, 0.2, 0.2, 1.0]
        self.specular = [0.0, 0.0, 0.0, 1.0]
        self.emissive = [0.0, 0.0, 0.0, 1.0]
        self.transparency = 1.0
        self.optical_density = 1.0
        self.shininess = 0.0
        self.texture = None
        self.vertex_format = ""
        self.chunks = []          # list of per-face float arrays, concatenated lazily
        self._flat = None

    @property
    def vertices(self):
        if self._flat is None:
            self._flat = (np.concatenate(self.chunks) if self.chunks
                          else np.zeros(0, dtype=np.float64))
        return self._flat

    @property
    def vertex_size(self):
        return {"T2F_N3F_V3F": 8, "T2F_V3F": 5, "N3F_V3F": 6, "V3F": 3}.get(self.vertex_format, 0)


def _parse_mtl(path, materials):
    cur = None
    with open(path, "r", errors="replace") as fh:
        for raw in fh:
            tok = raw.split("#", 1)[0].split()
            if not tok:
                continue
            key = tok[0]
            if key == "newmtl":
                cur = ObjMaterial(" ".join(tok[1:]))
                materials[cur.name] = cur
            elif cur is None:
                continue
            elif key == "Kd":
                cur.diffuse = [float(tok[1]), float(tok[2]), float(tok[3]), 1.0]
            elif key == "Ka":
                cur.ambient = [float(tok[1]), float(tok[2]), float(tok[3]), 1.0]
            elif key == "Ks":
                cur.specular = [float(tok[1]), float(tok[2]), float(tok[3]), 1.0]
            elif key == "Ke":
                cur.emissive = [float(tok[1]), float(tok[2]), float(tok[3]), 1.0]
            elif key == "d":
                cur.transparency = float(tok[1])
            elif key == "Tr":
                cur.transparency = 1.0 - float(tok[1])
            elif key == "Ni":
                cur.optical_density = float(tok[1])
            elif key == "Ns":
                cur.shininess = float(tok[1])
            # map_Kd etc.: the reference would call float() on a Texture object and fail
            # (Scene.py:86-87); textures on materials are out of scope.


class Wavefront:
    """``Wavefront(path).materials`` -> ordered dict name -> ObjMaterial."""

    def __init__(self, path, native=True):
        self.path = path
        self.materials = {}
        if native:
            self._parse_native()
        else:
            self._parse()

    _FORMATS = {0: "", 4: "V3F", 5: "T2F_V3F", 6: "N3F_V3F", 7: "T2F_N3F_V3F"}

    def _parse_native(self):
        import ctypes as C
        from . import _native
        L = _native.lib()
        h = C.c_void_p()
        _native.check(L.tirt_obj_load(os.fsencode(self.path), C.byref(h)))
        try:
            for i in range(L.tirt_obj_material_count(h)):
                name = C.create_string_buffer(1024)
                params = (C.c_double * 19)()
                fmt, isdef, nfl = C.c_int(), C.c_int(), C.c_longlong()
                _native.check(L.tirt_obj_material_info(h, i, name, 1024, params, C.byref(fmt), C.byref(isdef), C.byref(nfl)))
                m = ObjMaterial(name.value.decode(errors="replace"), bool(isdef.value))
                p = list(params)
                m.diffuse, m.ambient, m.specular, m.emissive = p[0:4], p[4:8], p[8:12], p[12:16]
                m.transparency, m.optical_density, m.shininess = p[16], p[17], p[18]
                m.vertex_format = self._FORMATS[fmt.value]
                flat = np.zeros(nfl.value, dtype=np.float64)
                _native.check(L.tirt_obj_material_vertices(h, i, flat, nfl.value))
                m._flat = flat
                self.materials[m.name] = m
        finally:
            L.tirt_obj_free(h)

    def parse(self):          # pywavefront API compatibility: parsing already happened
        return self

    def _parse(self):
        pos, nor, tex = [], [], []
        current = None
        base = os.path.dirname(self.path)
        with open(self.path, "r", errors="replace") as fh:
            for raw in fh:
                tok = raw.split("#", 1)[0].split()
                if not tok:
                    continue
                key = tok[0]
                if key == "v":
                    pos.append((float(tok[1]), float(tok[2]), float(tok[3])))
                elif key == "vn":
                    nor.append((float(tok[1]), float(tok[2]), float(tok[3])))
                elif key == "vt":
                    tex.append((float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0))
                elif key == "mtllib":
                    mtl = os.path.join(base, " ".join(tok[1:]))
                    if os.path.exists(mtl):
                        _parse_mtl(mtl, self.materials)
                elif key == "usemtl":
                    name = " ".join(tok[1:])
                    current = self.materials.get(name)
                    if current is None:
                        current = ObjMaterial(name, is_default=True)
                        self.materials[name] = current
                elif key == "f":
                    if current is None:
                        current = ObjMaterial("default%d" % len(self.materials), is_default=True)
                        self.materials[current.name] = current
                    self._face(current, tok[1:], pos, nor, tex)

    @staticmethod
    def _face(mat, toks, pos, nor, tex):
        corners = []
        has_vt = has_vn = False
        for t in toks:
            parts = t.split("/")
            vi = int(parts[0])
            ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
            ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
            has_vt |= ti != 0
            has_vn |= ni != 0
            corners.append((vi, ti, ni))
        if not mat.vertex_format:
            mat.vertex_format = ("T2F_" if has_vt else "") + ("N3F_" if has_vn else "") + "V3F"
        want_vt = mat.vertex_format.startswith("T2F")
        want_vn = "N3F" in mat.vertex_format

        def emit(c):
            vi, ti, ni = c
            out = []
            if want_vt:
                out.extend(tex[ti - 1 if ti > 0 else ti] if ti else (0.0, 0.0))
            if want_vn:
                out.extend(nor[ni - 1 if ni > 0 else ni] if ni else (0.0, 0.0, 0.0))
            out.extend(pos[vi - 1 if vi > 0 else vi])
            return out

        flat = []
        for k in range(2, len(corners)):
            flat.extend(emit(corners[0]))
            flat.extend(emit(corners[k - 1]))
            flat.extend(emit(corners[k]))
        if flat:
            mat.chunks.append(np.asarray(flat, dtype=np.float64))
            mat._flat = None
