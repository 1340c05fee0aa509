#!/bin/bash
# The bench lines of a round on ONE box (the last part of tools/prof_final.sh b, without the counter passes).  usage: bash tools/bench_lines.sh r03
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
QC=gpurun_out/quant_f8_1280.npz
timeout -k 10 500 python bench.py --profile-out gpurun_out/perop_${TAG}_f16.json > gpurun_out/${TAG}_f16.log 2>&1; tail -1 gpurun_out/${TAG}_f16.log | cut -c1-160
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --profile-out gpurun_out/perop_${TAG}_f32.json > gpurun_out/${TAG}_f32.log 2>&1; tail -1 gpurun_out/${TAG}_f32.log | cut -c1-160
timeout -k 10 300 python bench.py --dtype f8 --no-cpu-baseline --no-exact-f32 --profile-out gpurun_out/perop_${TAG}_f8_640.json > gpurun_out/${TAG}_f8_640.log 2>&1; tail -1 gpurun_out/${TAG}_f8_640.log | cut -c1-160
timeout -k 10 300 python bench.py --dtype f8 --imgsz 1280 --batch 16 --no-cpu-baseline --no-exact-f32 --profile-out gpurun_out/perop_${TAG}_f8_1280.json > gpurun_out/${TAG}_f8_1280.log 2>&1; tail -1 gpurun_out/${TAG}_f8_1280.log | cut -c1-160
timeout -k 10 200 python bench.py --workload classify --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/${TAG}_cls.log 2>&1; tail -1 gpurun_out/${TAG}_cls.log | cut -c1-160
timeout -k 10 120 python tools/bench_preprocess.py > gpurun_out/${TAG}_pre.log 2>&1; tail -1 gpurun_out/${TAG}_pre.log
