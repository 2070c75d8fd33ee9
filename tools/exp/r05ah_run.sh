cd $GRAFT_REPO_ROOT
for o in "bdpt_lanes=2" "bdpt_lanes=3" "bdpt_lanes=4" "bdpt_lanes=4 bdpt_batch_items=33554432" "bdpt_lanes=3 bdpt_batch_items=25165824"; do echo "== 64 spp $o"; for i in 1 2 3; do python tools/bdpt_bench.py 64 512 $o | cut -c40-100; done; done
for o in "bdpt_lanes=2" "bdpt_lanes=3 bdpt_batch_items=25165824" "bdpt_lanes=4 bdpt_batch_items=33554432"; do echo "== 256 spp $o"; for i in 1 2; do python tools/bdpt_bench.py 256 512 $o | cut -c40-100; done; done
