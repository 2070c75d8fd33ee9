"""tirt_lbvh_build is bit-identical to the oracle (which reproduces the reference's
nodelist.txt): Morton codes, sorted pairs (stable order), bvh_node, compact_node."""
import os

import numpy as np
import pytest

import oracle_api as oa
from common import duplicate_code_scene, tiny_scene
from ti_raytrace_amd import scenes, LBvh

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build_both(ex):
    ex.build_scene()
    o = oa.OracleScene(ex.scene, ex.cam)
    done = o.lbvh_build()
    assert done == ex.scene.primitive_count - 1
    return ex, o


def assert_same_lbvh(ex, o):
    n = ex.scene.primitive_count
    om, ob, oc = o.lbvh_get()
    gm, gb, gc = ex.scene.ctx.lbvh_download(n)
    assert np.array_equal(ex.scene.ctx.morton_download(n), o.morton_codes()), "unsorted Morton pairs"
    assert np.array_equal(gm, om), "sorted (code, prim) pairs / stable order"
    assert np.array_equal(gb.view(np.uint32), ob.view(np.uint32)), "bvh_node rows"
    assert np.array_equal(gc.view(np.uint32), oc.view(np.uint32)), "compact_node rows"


def test_cornell_matches_oracle_and_nodelist(gpu_ctx_ok):
    ex, o = build_both(scenes.cornell_box(32, 32, 4, device_id=0))
    assert_same_lbvh(ex, o)
    ref = open(os.path.join(GOLD, "nodelist.txt")).read()
    assert LBvh.format_nodelist(ex.scene.bvh.compact_node.to_numpy()) == ref


@pytest.mark.parametrize("ntri", [1, 2, 3, 7, 64, 257, 2049, 5000])
def test_small_and_ragged_sizes(gpu_ctx_ok, ntri):
    ex, o = build_both(tiny_scene(ntri, seed=ntri, device_id=0))
    assert_same_lbvh(ex, o)


def test_single_primitive_scene(gpu_ctx_ok):
    """n = 1: the root is a leaf (no internal node at all)."""
    from ti_raytrace_amd import Example, PT_RGB
    from ti_raytrace_amd import SceneData as SCD
    ex = Example.example(16, 16, 4, 0)
    mat = SCD.Material(); mat.type = SCD.MAT_DISNEY; mat.setRough(0.5); mat.setColor([0.8, 0.8, 0.8, 1.0]); mat.alebdoTex = -1
    ex.scene.add_mesh(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], dtype=np.float64), mat)
    ex.integrator = PT_RGB.PathTrace(16, 16, ex.cam, ex.scene, 64)
    ex.build_scene(); ex.frame_camera(0.8)
    o = oa.OracleScene(ex.scene, ex.cam)
    assert o.lbvh_build() == 0
    assert_same_lbvh(ex, o)


def test_duplicate_morton_codes(gpu_ctx_ok):
    ex, o = build_both(duplicate_code_scene(device_id=0))
    codes = o.lbvh_get()[0][:, 0]
    assert (np.diff(codes) == 0).sum() > 20            # the scene really has runs of equal codes
    assert_same_lbvh(ex, o)


def test_teapot(gpu_ctx_ok):
    ex = scenes.single_model(32, 32, 4, device_id=0)
    ex.scene.setup_data_cpu(); ex.integrator.setup_data_cpu(); ex.integrator.setup_data_gpu(); ex.scene.setup_data_gpu()
    o = oa.OracleScene(ex.scene, ex.cam)
    assert o.lbvh_build() == ex.scene.primitive_count - 1
    assert ex.scene.primitive_count == 25201
    assert_same_lbvh(ex, o)
    # process_normal (BVH point query) and total_area
    ex.scene.process_normal()
    o.process_normal(ex.scene.vertex_index_np)
    gv, ov = ex.scene.vertex.to_numpy(), o.vertex()
    same = (gv.view(np.uint32) == ov.view(np.uint32)) | (np.isnan(gv) & np.isnan(ov))
    assert same.all(), (~same).sum()
    ex.scene.total_area()
    assert ex.scene.light_area.to_numpy()[0] == np.float32(o.total_area())


def test_headline_100k(gpu_ctx_ok):
    ex, o = build_both(scenes.synthetic(64, 64, 4, device_id=0))
    assert ex.scene.primitive_count == 100001
    assert_same_lbvh(ex, o)
    st = ex.scene.ctx.stats()
    print("GPU LBVH build: %.3f ms for %d primitives" % (st["ms_build"], ex.scene.primitive_count))


@pytest.mark.parametrize("which", ["duplicates", "random700"])
def test_device_lbvh_equals_the_reference_text_lbvh(gpu_ctx_ok, which):
    """tests/golden/refkat_lbvh.npz: what accel/LBvh.py's own source text builds (executed as plain Python through the taichi stand-in of
    tools/refkat, build container only; tests/test_refkat.py has the details and holds the oracle to it)."""
    from common import refkat_lbvh_scene
    from test_refkat import GL
    ex = refkat_lbvh_scene(which, device_id=0)
    ex.build_scene()
    n = ex.scene.primitive_count
    gm, gb, gc = ex.scene.ctx.lbvh_download(n)
    assert np.array_equal(gm, GL["lbvh_%s_morton" % which])
    assert np.array_equal(gb.view(np.uint32), GL["lbvh_%s_bvh_node" % which].view(np.uint32))
    assert np.array_equal(gc.view(np.uint32), GL["lbvh_%s_compact_node" % which].view(np.uint32))


def test_device_smooth_normals_equal_the_reference_text(gpu_ctx_ok):
    """Scene.process_normal / total_area from the reference's source text (tests/golden/refkat_lbvh.npz, tests/test_refkat.py) vs k_smooth_normal / k_total_area."""
    from test_refkat import GL
    ex = scenes.single_model(16, 16, 4, model="sphere.obj", device_id=0)
    ex.build_scene()                      # (single_model.build_scene runs process_normal and total_area on the device)
    got = ex.scene.ctx.vertex_download(ex.scene.vertex_count)
    assert np.array_equal(got.view(np.uint32), GL["normals_vertex_after"].view(np.uint32))
    assert float(ex.scene.light_area.to_numpy()[0]) == float(GL["normals_total_area"][0])
