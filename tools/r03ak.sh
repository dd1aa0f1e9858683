#!/bin/bash
# the k = 20 MLP proof at the end of the round: kernel / copy / marker trace -> timeline, host trace, Gantt
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
CIRCUIT=mlp K=20 REPS=3 timeout 250 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d "$O/r03ak_prove" -- python "$R/tools/prove_bench.py" --pinned > "$O/r03ak_prove.log" 2>&1
DB=$(find "$O/r03ak_prove" -name '*.db' | head -1)
python "$R/tools/hosttrace.py" "$DB" 93 120 > "$O/r03ak_hosttrace.txt" 2>&1
python "$R/tools/timeline.py" "$DB" 93 > "$O/r03ak_timeline.txt" 2>&1
python "$R/tools/gantt.py" "$DB" 93 250 > "$O/r03ak_gantt.txt" 2>&1
rm -rf "$O/r03ak_prove"
head -7 "$O/r03ak_timeline.txt"; grep -A8 "^entry point" "$O/r03ak_hosttrace.txt"
