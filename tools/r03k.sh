#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_group.py tests/test_contexts_cpu.py -x -q > "$O/r03k_group.log" 2>&1; echo "group rc=$?" >> "$O/r03k_group.log"; tail -15 "$O/r03k_group.log"
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_group.py > "$O/r03k_pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/r03k_pytest.log"; tail -6 "$O/r03k_pytest.log"
cd /tmp
CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03k_prove.log" 2>&1
echo "$(tail -1 $O/r03k_prove.log | grep -o '"prove_seconds_gpu_runs": [^]]*]') $(tail -1 $O/r03k_prove.log | grep -o '"proof_sha256": "[0-9a-f]*"')"
