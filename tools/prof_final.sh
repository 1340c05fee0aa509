cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1 && echo stats-ok && cd $R && bash tools/pmc_profile.sh final > gpurun_out/pmc_final.log 2>&1; tail -3 gpurun_out/pmc_final.log
