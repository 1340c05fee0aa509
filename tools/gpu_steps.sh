#!/bin/bash
# Run GPU steps one after another on the gpurun box; a step that times out (or is killed) ends the whole call - no further
# GPU step is started behind a hung one - while an ordinary failure (a failing test) only gets reported.
#   usage: source tools/gpu_steps.sh; step SECONDS LOGFILE command...
mkdir -p gpurun_out
step() {
  local secs=$1 log=$2; shift 2
  echo "[step] $* -> $log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "[step] rc=$rc $log: $(tail -n 1 "$log" | cut -c1-200)"
  # 124 / 137: timed out / killed; >= 128: ended by a signal (134 = abort after a GPU memory access fault): no further GPU step
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "[step] timed out / killed / aborted: stopping"; exit $rc; fi
  return 0
}
