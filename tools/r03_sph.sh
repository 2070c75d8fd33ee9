R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_spectral.py -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do timeout 600 python bench.py --configs-only spectral_cornell_512x512_64spp 2>&1 | tail -1 | cut -c1-200; done
