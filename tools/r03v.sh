#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
CIRCUIT=mlp K=20 REPS=3 timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d "$O/r03v_prove" -- python "$R/tools/prove_bench.py" --pinned > "$O/r03v_prove.log" 2>&1
DB=$(find "$O/r03v_prove" -name '*.db' | head -1)
python "$R/tools/gantt.py" "$DB" 95 250 > "$O/r03v_prove_gantt.txt" 2>&1
python "$R/tools/hosttrace.py" "$DB" 95 120 > "$O/r03v_prove_hosttrace.txt" 2>&1
python "$R/tools/timeline.py" "$DB" 95 > "$O/r03v_prove_timeline.txt" 2>&1
rm -rf "$O/r03v_prove"
head -8 "$O/r03v_prove_timeline.txt"; cat "$O/r03v_prove_hosttrace.txt" | head -150
