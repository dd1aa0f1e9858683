#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$O/r03b_msmcols" -- python "$R/tools/msm_columns_profile.py" run > "$O/r03b_msmcols.log" 2>&1
DB=$(find "$O/r03b_msmcols" -name '*.db' | head -1)
python "$R/tools/msm_columns_profile.py" reduce "$DB" "$O/r03b_msmcols.log" > "$O/r03b_msmcols_table.txt" 2>&1
rm -rf "$O/r03b_msmcols"
CIRCUIT=mlp K=20 REPS=3 timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d "$O/r03b_prove" -- python "$R/tools/prove_bench.py" --pinned > "$O/r03b_prove.log" 2>&1
DB=$(find "$O/r03b_prove" -name '*.db' | head -1)
python "$R/tools/gantt.py" "$DB" 105 250 > "$O/r03b_prove_gantt.txt" 2>&1
python "$R/tools/hosttrace.py" "$DB" 105 100 > "$O/r03b_prove_hosttrace.txt" 2>&1
rm -rf "$O/r03b_prove"
CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03b_prove_plain.log" 2>&1
cat "$O/r03b_msmcols_table.txt"; tail -1 "$O/r03b_prove_plain.log" | cut -c1-1500
