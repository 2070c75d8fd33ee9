cd $GRAFT_REPO_ROOT
for g in 0 512 256 128 64; do echo "== grid $g"; for i in 1 2 3; do TIRT_PVB_GRID=$g python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 20 --warmup 5 | tail -1 | cut -c90-170; done; done
