#!/bin/bash
# Round-end evidence run (on the GPU box, via gpurun): rocprofv3 kernel stats of the default bench, the PMC passes,
# the bench lines (f16 default incl. parity / exact_f32 / CPU baseline, fp8 at 640 and at config 5's 1280x1280x16,
# classifier) and the device pre-processing rates.   usage: bash tools/prof_final.sh <tag e.g. r02>
TAG=${1:-r03}
PHASE=${2:-all}      # a: kernel stats + f16 counters, b: fp8 counters + bench lines (gpurun calls are limited to 20 minutes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
QC=$R/gpurun_out/quant_f8_1280.npz
if [ "$PHASE" != "b" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-exact-f32 --no-classify > $R/gpurun_out/prof_$TAG.log 2>&1 && echo stats-ok
# the same command with the Detect head's chains kept on the caller's stream: kernels do not overlap, so per-kernel average
# durations are comparable with bench.py's per-op event timing (its roofline block runs in that mode)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_inorder -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-exact-f32 --no-classify --opt head_lanes=0 > $R/gpurun_out/prof_${TAG}_inorder.log 2>&1 && echo stats-inorder-ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_cls -- python3 $R/bench.py --workload classify --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_${TAG}_cls.log 2>&1 && echo stats-cls-ok
# fp8: one unprofiled run first writes the calibration (QuantSpec) to a cache, so that the profiled runs hold the fp8 step only
rm -f $QC
(cd $R && timeout -k 10 300 python3 bench.py --dtype f8 --imgsz 1280 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-exact-f32 --no-classify --no-roofline --quant-cache $QC > gpurun_out/${TAG}_f8_calib.log 2>&1) && echo calib-f8-ok
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_f8 -- python3 $R/bench.py --dtype f8 --imgsz 1280 --batch 16 --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-exact-f32 --no-classify --quant-cache $QC > $R/gpurun_out/prof_${TAG}_f8.log 2>&1 && echo stats-f8-ok
cd $R && bash tools/pmc_profile.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; tail -2 gpurun_out/pmc_$TAG.log
fi
if [ "$PHASE" != "a" ]; then
cd $R
# (each gpurun call lands on a fresh box: write the fp8 calibration cache again, unprofiled, before the counter passes)
rm -f $QC
timeout -k 10 300 python3 bench.py --dtype f8 --imgsz 1280 --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-exact-f32 --no-classify --no-roofline --quant-cache $QC > gpurun_out/${TAG}_f8_calib.log 2>&1 && echo calib-f8-ok
bash tools/pmc_profile.sh ${TAG}_f8 --dtype f8 --imgsz 1280 --batch 16 --quant-cache $QC > gpurun_out/pmc_${TAG}_f8.log 2>&1; tail -1 gpurun_out/pmc_${TAG}_f8.log
timeout -k 10 500 python bench.py --profile-out gpurun_out/perop_${TAG}_f16.json > gpurun_out/${TAG}_f16.log 2>&1; tail -1 gpurun_out/${TAG}_f16.log | cut -c1-160
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --profile-out gpurun_out/perop_${TAG}_f32.json > gpurun_out/${TAG}_f32.log 2>&1; tail -1 gpurun_out/${TAG}_f32.log | cut -c1-160
timeout -k 10 300 python bench.py --dtype f8 --no-cpu-baseline --no-exact-f32 --profile-out gpurun_out/perop_${TAG}_f8_640.json > gpurun_out/${TAG}_f8_640.log 2>&1; tail -1 gpurun_out/${TAG}_f8_640.log | cut -c1-160
timeout -k 10 300 python bench.py --dtype f8 --imgsz 1280 --batch 16 --no-cpu-baseline --no-exact-f32 --profile-out gpurun_out/perop_${TAG}_f8_1280.json > gpurun_out/${TAG}_f8_1280.log 2>&1; tail -1 gpurun_out/${TAG}_f8_1280.log | cut -c1-160
timeout -k 10 200 python bench.py --workload classify --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/${TAG}_cls.log 2>&1; tail -1 gpurun_out/${TAG}_cls.log | cut -c1-160
timeout -k 10 120 python tools/bench_preprocess.py > gpurun_out/${TAG}_pre.log 2>&1; tail -1 gpurun_out/${TAG}_pre.log
fi
