R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03sph_pytest.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/r03sph_pytest.log | tail -8
