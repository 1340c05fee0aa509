#!/bin/bash
# Hardware-counter passes over a short bench run (run ON the GPU box, via gpurun):
#   bash tools/pmc_profile.sh <tag> [bench args...]
# One rocprofv3 invocation per counter group (PMC slots are limited; FETCH_SIZE/WRITE_SIZE need
# passes of their own), each with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
set -uo pipefail
TAG=${1:-pmc}; shift || true
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A G
G[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES"
G[sq2]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
G[tcp]="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
for g in ${PMC_GROUPS:-sq1 sq2 tcc fetch write tcp}; do
  timeout -k 10 240 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d "$OUT/$g" -- \
    python3 "$R/bench.py" --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-parity --no-exact-f32 --no-classify "$@" > "$OUT/$g.log" 2>&1
  echo "pass $g rc=$? $(ls $OUT/$g/*/ 2>/dev/null | tr '\n' ' ')"
done
