#!/bin/bash
# Build libmiyolo.so for gfx950 (hipcc cross-compiles without a GPU).
#   -ffp-contract=off : the NMS box arithmetic must round exactly like the CPU reference
#                       (no FMA contraction); FMAs that are wanted are written as fmaf().
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
EXTRA=()
if [ "${1:-}" = "ablate" ]; then EXTRA=(-DMIYOLO_ABLATE=1); shift; fi   # timing-experiment build, never shipped
if [ "${1:-}" = "experiments" ]; then EXTRA=(-DMIYOLO_EXPERIMENTS=1); shift; fi   # + conv_halo/halop/ws/dmh (conv_impl 2,4,5,6)
if [ "${1:-}" = "stamps" ]; then EXTRA=(-DMIYOLO_ABLATE=2); shift; fi   # in-kernel cycle stamps only, never shipped
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
  -o "${OUT:-libmiyolo.so}" miyolo.hip "${EXTRA[@]}" "$@"
echo "built $(pwd)/${OUT:-libmiyolo.so}"
