#!/bin/bash
# the two PMC passes (separately) over the prove workload -> gpurun_out/<tag>_pmc_prove.json (copy into profiles/ to have it judged)
set -e
TAG=${1:-pmc}
R=$(cd "$(dirname "$0")/.." && pwd)
O="$R/gpurun_out/${TAG}_pmcp"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python "$R/tools/pmc_prove.py" > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python "$R/tools/pmc_prove.py" > "$O/write.log" 2>&1
F=$(find "$O/fetch" -name '*counter_collection.csv' | head -1); W=$(find "$O/write" -name '*counter_collection.csv' | head -1)
python "$R/tools/pmc_prove_reduce.py" "$F" "$W" "$R/gpurun_out/${TAG}_pmc_prove.json" "$TAG" | tail -20
rm -rf "$O"
