#!/bin/bash
# rocprofv3 kernel trace of one native k = 20 proof (tools/prove_bench.py --native) -> gpurun_out/<tag>_prove_kernel_stats.txt
set -e
TAG=${1:-prove}
R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p "$R/gpurun_out/${TAG}_prove"
cd /tmp && export TMPDIR=/tmp
K=${K:-20} BLOCKS=${BLOCKS:-4} rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/${TAG}_prove" -- python "$R/tools/prove_bench.py" --native > "$R/gpurun_out/${TAG}_prove.log" 2>&1 || true
DB=$(find "$R/gpurun_out/${TAG}_prove" -name '*.db' | head -1)
python "$R/tools/rocpd_stats.py" "$DB" > "$R/gpurun_out/${TAG}_prove_kernel_stats.txt"
head -40 "$R/gpurun_out/${TAG}_prove_kernel_stats.txt"
