#!/bin/bash
# round-3 first measurement pass: GPU tests, serial per-column MSM kernel profile, k = 20 MLP proof timeline
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/r03a_pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/r03a_pytest.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$O/r03a_msmcols" -- python "$R/tools/msm_columns_profile.py" run > "$O/r03a_msmcols.log" 2>&1
DB=$(find "$O/r03a_msmcols" -name '*.db' | head -1)
python "$R/tools/msm_columns_profile.py" reduce "$DB" "$O/r03a_msmcols.log" > "$O/r03a_msmcols_table.txt" 2>&1
rm -rf "$O/r03a_msmcols"
CIRCUIT=mlp K=20 REPS=3 timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d "$O/r03a_prove" -- python "$R/tools/prove_bench.py" --pinned > "$O/r03a_prove.log" 2>&1
DB=$(find "$O/r03a_prove" -name '*.db' | head -1)
python "$R/tools/gantt.py" "$DB" 110 250 > "$O/r03a_prove_gantt.txt" 2>&1
python "$R/tools/timeline.py" "$DB" 110 > "$O/r03a_prove_timeline.txt" 2>&1
rm -rf "$O/r03a_prove"
CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03a_prove_plain.log" 2>&1
cd "$R" && timeout 300 python bench.py --no-cpu-baseline > "$O/r03a_bench.log" 2>&1
tail -3 "$O/r03a_pytest.log"; cat "$O/r03a_msmcols_table.txt"; tail -1 "$O/r03a_prove_plain.log" | cut -c1-1500
