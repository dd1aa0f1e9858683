#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=1 timeout 1500 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d "$O/r03p_k22" -- python "$R/tools/prove_bench.py" --pinned > "$O/r03p_k22.log" 2>&1
DB=$(find "$O/r03p_k22" -name '*.db' | head -1)
python "$R/tools/gantt.py" "$DB" 1600 5000 > "$O/r03p_k22_gantt.txt" 2>&1
python "$R/tools/hosttrace.py" "$DB" 1600 3000 > "$O/r03p_k22_hosttrace.txt" 2>&1
python "$R/tools/timeline.py" "$DB" 1600 > "$O/r03p_k22_timeline.txt" 2>&1
rm -rf "$O/r03p_k22"
cut -c1-340 "$O/r03p_k22_gantt.txt"; head -60 "$O/r03p_k22_hosttrace.txt"; head -30 "$O/r03p_k22_timeline.txt"
