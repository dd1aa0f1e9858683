#!/bin/bash
# rocprofv3 kernel trace of `python bench.py` -> gpurun_out/<tag>_kernel_stats.txt (copy into profiles/ to have it judged)
#   usage (on the GPU box): bash tools/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-prof}; shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$REPO/gpurun_out/$TAG"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/$TAG" -- python "$REPO/bench.py" --no-cpu-baseline --no-side-legs --steps 20 --warmup 5 "$@" > "$REPO/gpurun_out/${TAG}_bench_profiled.log" 2>&1 || true
DB=$(find "$REPO/gpurun_out/$TAG" -name '*.db' | head -1)
python "$REPO/tools/rocpd_stats.py" "$DB" > "$REPO/gpurun_out/${TAG}_kernel_stats.txt"
grep "^{\"metric" "$REPO/gpurun_out/${TAG}_bench_profiled.log" | cut -c1-400
head -24 "$REPO/gpurun_out/${TAG}_kernel_stats.txt"
