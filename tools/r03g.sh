#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/r03g_pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/r03g_pytest.log"
tail -30 "$O/r03g_pytest.log"
cd /tmp && export TMPDIR=/tmp
CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03g_prove.log" 2>&1
echo "$(tail -1 $O/r03g_prove.log | grep -o '"prove_seconds_gpu_runs": [^]]*]') $(tail -1 $O/r03g_prove.log | grep -o '"prove_breakdown_seconds": {[^}]*}') $(tail -1 $O/r03g_prove.log | grep -o '"proof_sha256": "[0-9a-f]*"')"
tail -3 "$O/r03g_prove.log" | cut -c1-600
