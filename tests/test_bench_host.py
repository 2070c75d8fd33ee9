"""Host-side helpers of bench.py that need no GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_host_cpu_info_describes_the_box():
    import bench
    info = bench.host_cpu_info()
    assert info["nproc"] >= 1 and (info["affinity"] is None or info["affinity"] >= 1)
    assert info["cgroup_cpu_quota"] is None or info["cgroup_cpu_quota"] > 0
    assert "cpu_model" in info


def test_native_oracle_build_equals_the_portable_one():
    """bench.py times a -O3 -march=native build of the oracle made on the box it runs on; with contraction and fast-math off it
    must compute the same film as the portable build the tests use."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as oa
    from common import host_only
    from ti_raytrace_amd import scenes
    if oa.build_native() is None:
        import pytest
        pytest.skip("no compiler for the native oracle build")
    ex = host_only(scenes.synthetic(48, 48, 4, ntri=2000))
    a = oa.OracleScene(ex.scene, ex.cam, native=True); a.lbvh_build()
    b = oa.OracleScene(ex.scene, ex.cam); b.lbvh_build()
    fa, sa = a.render(48, 48, 0, 2, seed=1, nthreads=4)
    fb, sb = b.render(48, 48, 0, 2, seed=1, nthreads=4)
    assert np.array_equal(fa, fb) and sa == sb
