#!/bin/bash
# tools/prof_steps.sh <tag> [ENV=...]: rocprofv3 kernel trace of tools/msm_steps.py -> gpurun_out/<tag>_steps_kernel_stats.txt
TAG=${1:?tag}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$REPO/gpurun_out/$TAG"
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/$TAG" -- python "$REPO/tools/msm_steps.py" > "$REPO/gpurun_out/${TAG}_steps.log" 2>&1 || true
DB=$(find "$REPO/gpurun_out/$TAG" -name '*.db' | head -1)
python "$REPO/tools/rocpd_stats.py" "$DB" > "$REPO/gpurun_out/${TAG}_steps_kernel_stats.txt"
rm -rf "$REPO/gpurun_out/$TAG"
grep "wall" "$REPO/gpurun_out/${TAG}_steps.log"; head -28 "$REPO/gpurun_out/${TAG}_steps_kernel_stats.txt"
