#!/bin/bash
# Round-end evidence run (on the GPU box, via gpurun): rocprofv3 kernel stats of the default bench, the PMC passes,
# the final bench lines (f16 with CPU baseline, f32, classifier) and the device pre-processing rates.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1 && echo stats-ok
cd $R && bash tools/pmc_profile.sh final > gpurun_out/pmc_final.log 2>&1; tail -2 gpurun_out/pmc_final.log
timeout -k 10 400 python bench.py --profile-out gpurun_out/perop_final_f16.json > gpurun_out/final_f16.log 2>&1; tail -1 gpurun_out/final_f16.log | cut -c1-160
timeout -k 10 300 python bench.py --dtype f32 --no-cpu-baseline --profile-out gpurun_out/perop_final_f32.json > gpurun_out/final_f32.log 2>&1; tail -1 gpurun_out/final_f32.log | cut -c1-160
timeout -k 10 200 python bench.py --workload classify --no-cpu-baseline > gpurun_out/final_cls.log 2>&1; tail -1 gpurun_out/final_cls.log | cut -c1-160
timeout -k 10 120 python tools/bench_preprocess.py > gpurun_out/final_pre.log 2>&1; tail -1 gpurun_out/final_pre.log
