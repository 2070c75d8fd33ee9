R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_trace.py -q -m gpu -x -k "timeline or measurement" -s 2>&1 | grep -E "timeline:|passed|failed|Error" | tail -5
timeout 600 python bench.py --configs-only spectral_cornell_512x512_64spp 2>&1 | tail -1 | cut -c1-900
