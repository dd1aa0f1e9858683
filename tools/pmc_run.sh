#!/bin/bash
# the two PMC passes (separately, as the guide requires) -> gpurun_out/<tag>_pmc/{FETCH,WRITE}.csv + gpurun_out/<tag>_pmc_traffic.json
set -e
TAG=${1:-pmc}
R=$(cd "$(dirname "$0")/.." && pwd)
O="$R/gpurun_out/${TAG}_pmc"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python "$R/tools/pmc_probe.py" > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python "$R/tools/pmc_probe.py" > "$O/write.log" 2>&1
F=$(find "$O/fetch" -name '*counter_collection.csv' | head -1); W=$(find "$O/write" -name '*counter_collection.csv' | head -1)
cp "$F" "$O/FETCH_SIZE_counter_collection.csv"; cp "$W" "$O/WRITE_SIZE_counter_collection.csv"
python "$R/tools/pmc_reduce.py" "$F" "$W" "$R/gpurun_out/${TAG}_pmc_traffic.json" "$TAG" | tail -25
rm -rf "$O/fetch" "$O/write"
