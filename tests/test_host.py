"""Host logic (no device): OBJ/MTL ingest, struct packing, camera, texture, scene generators."""
import numpy as np
import pytest

from common import host_only
from ti_raytrace_amd import scenes, Camera, Texture, ObjLoader
from ti_raytrace_amd import SceneData as SCD


def test_cornell_ingest_order_and_classification():
    ex = host_only(scenes.cornell_box(32, 32, 4))
    s = ex.scene
    assert s.primitive_count == 36 and s.vertex_count == 108 and s.material_count == 4
    assert s.material_np[:, 0].tolist() == [0.0, 0.0, 0.0, 2.0]          # white red green (Disney) light
    assert s.material_np[3, 2:5].tolist() == [10.0, 10.0, 10.0]
    assert s.material_np[0, 5:7].tolist() == [0.0, 0.5]
    assert s.light_np.tolist() == [34, 35] and s.light_count == 2
    assert (np.diff(s.primitive_np[:, 2]) >= 0).all()                   # grouped by material in MTL order
    assert s.primitive_np[:, 1].tolist() == list(range(0, 108, 3))
    assert s.minboundarynp.tolist() == [[0.0, 0.0, np.float32(-559.2)]]
    assert s.maxboundarynp.tolist() == [[556.0, np.float32(548.8), 0.0]]
    # face normals were generated and are unit length
    n = s.vertex_np[:, 3:6]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
    assert s.vertex_index_np.tolist() == [i // 3 for i in range(108)]


def test_obj_formats_and_fan(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0.5 0.25\n"
                 "f 1/1/1 2/1/1 3/1/1 4/1/1\nusemtl other\nf 1 2 3\n")
    w = ObjLoader.Wavefront(str(p))
    names = list(w.materials)
    assert names == ["default0", "other"]
    a, b = w.materials["default0"], w.materials["other"]
    assert a.vertex_format == "T2F_N3F_V3F" and b.vertex_format == "V3F"
    va = a.vertices.reshape(-1, 8)
    assert va.shape[0] == 6                                             # quad -> 2 triangles
    assert va[:, 5:8].tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 0, 0], [1, 1, 0], [0, 1, 0]]
    assert b.vertices.reshape(-1, 3).shape[0] == 3
    assert a.transparency == 1.0 and a.diffuse[:3] == [0.8, 0.8, 0.8]


def _same_wavefront(path):
    a, b = ObjLoader.Wavefront(str(path), native=True), ObjLoader.Wavefront(str(path), native=False)
    assert list(a.materials) == list(b.materials)
    for name in a.materials:
        x, y = a.materials[name], b.materials[name]
        assert x.vertex_format == y.vertex_format and x.is_default == y.is_default, name
        for f in ("diffuse", "ambient", "specular", "emissive"):
            assert list(getattr(x, f)) == list(getattr(y, f)), (name, f)
        assert (x.transparency, x.optical_density, x.shininess) == (y.transparency, y.optical_density, y.shininess)
        assert x.vertices.dtype == np.float64 and np.array_equal(x.vertices.view(np.uint64), y.vertices.view(np.uint64)), name
    return a


def test_native_obj_reader_equals_the_python_one_on_the_shipped_models():
    """csrc/tirt_obj.hip (strtod, same grouping rules) against the pure-Python restatement of PyWavefront:
    identical material order, parameters and vertex doubles on every model under assets/."""
    import glob, os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "model")
    files = sorted(glob.glob(os.path.join(root, "*.obj")))
    assert len(files) >= 5
    for f in files:
        w = _same_wavefront(f)
        assert sum(m.vertices.size for m in w.materials.values()) > 0, f


def test_native_obj_reader_edge_cases(tmp_path):
    (tmp_path / "m.mtl").write_text("# comment\nnewmtl red\nKd 1 0 0\nKe 0 0 0\nNi 1.5\nNs 32\nd 0.25\n\nnewmtl lamp\r\nKe 17 12 4 # warm\nTr 0.1\n"
                                    "newmtl red\nKd 0.5 0.25 0.125\n")
    (tmp_path / "m.obj").write_text(
        "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 0.5 1e-3\nvn 0 0 1\nvn 0 1 0\nvt 0.5 0.25\nvt 0.75\n"
        "f 1 2 3\n"                               # before any usemtl: default material
        "usemtl red\nf 1//1 2//2 3//1 4//2 5//1\n"   # pentagon, N3F_V3F
        "f -1//-1 -2//-2 -3//1\r\n"                 # negative indices, CRLF
        "usemtl lamp\nf 1/1 2/2 3/1\n"
        "usemtl unknown\nf 1/2/1 2/1/2 3/2/1 # trailing comment\n"
        "usemtl red\nf 3 4 5\n"                      # format of `red` stays N3F_V3F: absent normals become zeros
        "g ignored\no also\ns off\nf 1 2\n")        # two corners: nothing emitted
    w = _same_wavefront(tmp_path / "m.obj")
    assert list(w.materials) == ["red", "lamp", "default2", "unknown"]
    assert w.materials["red"].diffuse[:3] == [0.5, 0.25, 0.125] and w.materials["red"].optical_density == 1.0   # redefinition replaces
    assert w.materials["lamp"].emissive[:3] == [17.0, 12.0, 4.0] and abs(w.materials["lamp"].transparency - 0.9) < 1e-15
    assert w.materials["red"].vertices.size == (3 + 1 + 1) * 3 * 6 and w.materials["unknown"].vertex_format == "T2F_N3F_V3F"
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(Exception):
        ObjLoader.Wavefront(str(bad))
    with pytest.raises(Exception):
        ObjLoader.Wavefront(str(tmp_path / "missing.obj"))


def test_struct_packing_roundtrip():
    m = SCD.Material(); m.type = SCD.MAT_GLASS; m.alebdoTex = -1; m.setColor([0.1, 0.2, 0.3]); m.setIor(1.3); m.setExtinciton(5.0)
    buf = np.zeros((2, SCD.MAT_VEC_SIZE), np.float32); m.fillStruct(buf, 1)
    assert np.allclose(buf[1], [1, -1, 0.1, 0.2, 0.3, 1.3, 5.0, 0, 0, 0])
    sh = SCD.Shape(); sh.type = SCD.SHPAE_SPHERE; sh.pos = [1, 2, 3]; sh.setRadius(4)
    buf = np.zeros((1, SCD.SHA_VEC_SIZE), np.float32); sh.fillStruct(buf, 0)
    assert buf[0].tolist() == [1, 1, 2, 3, 4, 0, 0, 0, 0, 0]
    pr = SCD.Primitive(); pr.type = SCD.PRIMITIVE_SHAPE; pr.vertex_shape_index = 7; pr.mat_index = 2
    buf = np.zeros((1, 3), np.int32); pr.fillStruct(buf, 0)
    assert buf[0].tolist() == [2, 7, 2]


def test_camera_matrices():
    cam = Camera.Camera(64, 48, 16)
    assert cam.fx == 2.0 * 64 / 2.4 and cam.fy == cam.fx and cam.cx == 32 and cam.cy == 24
    cam.scale = 10.0
    cam.set_target(1.0, 2.0, 3.0)
    assert np.allclose(cam.eye_np[0], [1, 2, 13])
    v, vi = cam.view_np[0], cam.view_inv_np[0]
    assert v.dtype == np.float32 and vi.dtype == np.float32
    assert np.allclose(v @ vi, np.eye(4), atol=1e-5)
    assert np.allclose(v[:3, :3], np.eye(3))                               # yaw = pitch = 0 looks down -z
    cam.update_frame(); cam.update_frame(3)
    assert cam.frame == 4 and cam.frame_cpu[0] == 4
    Camera.Camera(8, 8, 1)                                                  # spp < 4 must not raise (quirk B6 not reproduced)


def test_texture_packing():
    t = Texture.Texture()
    img = np.zeros((2, 3, 3), np.int32)
    img[0, 1] = (10, 20, 30)            # top row, x = 1
    t.load_array(img)
    assert t.wid == 3 and t.hgt == 2 and t.np_img.shape == (3, 2)
    assert t.np_img[1, 1] == (10 << 16) | (20 << 8) | 30                    # y flipped: top row -> y = hgt-1
    assert t.np_img.sum() == t.np_img[1, 1]


def test_synthetic_scene_is_reproducible():
    a = scenes.synthetic_triangles(1000, 1234)
    b = scenes.synthetic_triangles(1000, 1234)
    assert np.array_equal(a, b) and a.shape == (1000, 3, 3)
    c = a.mean(axis=1)
    assert np.abs(c).max() <= 1.03 and np.abs(a - c[:, None, :]).max() <= 0.06
    u = scenes.splitmix64_unit(1234, 4)
    assert (u >= 0).all() and (u < 1).all() and len(set(u.tolist())) == 4
    ex = host_only(scenes.synthetic(16, 16, 4, ntri=50))
    assert ex.scene.primitive_count == 51 and ex.scene.light_np.tolist() == [50]
    assert ex.scene.primitive_np[50].tolist() == [2, 0, 1]


def test_default_tile_size():
    """PT_RGB.default_tile_size: 8 whole columns where that is between 4096 and 16384 pixels (the device then walks a tile in 8 x 8
    pixel blocks), 4096 linear pixels otherwise."""
    from ti_raytrace_amd.PT_RGB import default_tile_size
    assert default_tile_size(1024) == 8192 and default_tile_size(512) == 4096 and default_tile_size(2048) == 16384
    assert default_tile_size(4096) == 4096 and default_tile_size(100) == 4096 and default_tile_size(516) == 4096


def test_bench_big_scenes_fit_the_reference_node_format():
    """compact_node keeps node indices as f32 (SceneData.py / UtilsFunc.py: exact below 2^24), as the reference does: 2 n - 1 nodes for n primitives, so
    bench.py's big-scene configs (triangles + the sphere light) must stay below 2^23 primitives -- 16 M triangles would not build (tirt_scene_upload refuses them)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert bench.BIG_SCENES, "no big-scene configs"
    for name, (ntri, spread, npx) in bench.BIG_SCENES.items():
        assert 2 * (ntri + 1) - 1 < 2 ** 24, name
        assert ("%dM" % (ntri // 1000000)) in name and npx >= 256 and 0.0 < spread < 0.03
