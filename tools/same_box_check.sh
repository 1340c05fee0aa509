#!/bin/bash
# rocprofv3's per-kernel average and bench.py's own HIP-event average of the dominant kernel ON ONE BOX (boxes differ by a few
# percent in clock; the judged numbers should be compared box for box).  usage (GPU box): bash tools/same_box_check.sh r03
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/samebox_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-exact-f32 --no-classify --no-roofline --opt head_lanes=0 > $R/gpurun_out/samebox_${TAG}_prof.log 2>&1 && echo stats-ok
cd $R && timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-exact-f32 --no-classify > gpurun_out/samebox_${TAG}_bench.log 2>&1 && echo bench-ok
