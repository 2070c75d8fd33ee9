cd $GRAFT_REPO_ROOT
run() { echo "== $*"; for i in 1 2; do python bench.py --no-cpu-baseline --no-roofline --no-configs --steps 20 --warmup 5 "$@" | tail -1 | cut -c90-170; done; }
run
M=1048576
run --opt overlap_lanes=2 --opt batch_paths=$((107*M)) --opt merge_paths=$((107*M))
run --opt overlap_lanes=3 --opt batch_paths=$((107*M)) --opt merge_paths=$((107*M))
run --opt overlap_lanes=3 --opt batch_paths=$((128*M)) --opt merge_paths=$((128*M))
run --opt overlap_lanes=4 --opt batch_paths=$((80*M)) --opt merge_paths=$((80*M))
run --opt overlap_lanes=3 --opt batch_paths=$((64*M)) --opt merge_paths=$((64*M))
run --opt overlap_lanes=2 --opt batch_paths=$((64*M)) --opt merge_paths=$((64*M))
