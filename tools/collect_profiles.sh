#!/bin/bash
# Copy the judged evidence of a round from gpurun_out/ (scratch) into profiles/ (committed).  usage: bash tools/collect_profiles.sh r02
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
newest() { ls -t "$1"/*/*_kernel_stats.csv | head -1; }
cp "$(newest gpurun_out/prof_$TAG)"          profiles/${TAG}_kernel_stats_f16_b64.csv
cp "$(newest gpurun_out/prof_${TAG}_inorder)" profiles/${TAG}_kernel_stats_inorder.csv
cp "$(newest gpurun_out/prof_${TAG}_cls)"     profiles/${TAG}_kernel_stats_cls_b256.csv
cp "$(newest gpurun_out/prof_${TAG}_f8)"      profiles/${TAG}_kernel_stats_f8_1280_b16.csv
python tools/pmc_summarize.py gpurun_out/pmc_$TAG profiles/${TAG}_pmc_f16_b64.md > /dev/null
python tools/pmc_summarize.py gpurun_out/pmc_${TAG}_f8 profiles/${TAG}_pmc_f8_1280_b16.md > /dev/null
python tools/pmc_traffic.py gpurun_out/pmc_$TAG profiles/${TAG}_traffic.json 64 640 640 f16
python tools/pmc_traffic.py gpurun_out/pmc_${TAG}_f8 profiles/${TAG}_traffic_f8_1280_b16.json 16 1280 1280 f8
python tools/per_layer_table.py gpurun_out/perop_${TAG}_f16.json f16 profiles/${TAG}_per_layer_f16.md
python tools/per_layer_table.py gpurun_out/perop_${TAG}_f32.json f32 profiles/${TAG}_per_layer_f32.md
python tools/per_layer_table.py gpurun_out/perop_${TAG}_f8_1280.json f8 profiles/${TAG}_per_layer_f8_1280.md || true
{
  echo "# Bench lines of round ${TAG#r} (one MI355X, tools/prof_final.sh; each is the single JSON line bench.py prints)"
  for f in f16 f32 f8_640 f8_1280 cls; do
    echo; echo "## $f"; echo '```json'; tail -1 gpurun_out/${TAG}_$f.log; echo '```'
  done
  echo; echo "## device pre-processing (tools/bench_preprocess.py)"; echo '```'; grep '^{' gpurun_out/${TAG}_pre.log; echo '```'
} > profiles/${TAG}_bench_lines.md
[ -f gpurun_out/${TAG/r0/r}_stamps_h2.log ] && grep -v amdgpu.ids gpurun_out/${TAG/r0/r}_stamps_h2.log > profiles/${TAG}_h2_stamps.log
ls -la profiles | grep ${TAG}_
