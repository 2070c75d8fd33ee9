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


def test_gpus_n_without_a_launcher_starts_one_process_per_gpu(monkeypatch):
    """`python bench.py --gpus 4 --steps 2` as the driver types it: no WORLD_SIZE in the environment -> the script re-executes
    itself under torch.distributed.run (127.0.0.1, a free port, the same arguments) before torch or HIP are touched."""
    import subprocess
    import pytest
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7                                   # the launcher's exit code is the script's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert os.path.samefile(cmd[cmd.index("--master-port") + 2], os.path.join(ROOT, "bench.py"))
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
