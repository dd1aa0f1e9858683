#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_contexts_cpu.py tests/test_execute.py tests/test_group.py -x -q > "$O/r03m_pytest.log" 2>&1; echo "rc=$?" >> "$O/r03m_pytest.log"; tail -5 "$O/r03m_pytest.log"
cd /tmp
for C in 0,0 0,0,0,0; do
CONTEXTS=$C CIRCUIT=mlp K=20 REPS=3 timeout 900 python "$R/tools/prove_group.py" --pinned > "$O/r03m_group_$C.log" 2>&1; tail -1 "$O/r03m_group_$C.log" | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['contexts'], 'same', j['same_bytes_as_one_context'], 'verifies', j['verifier_accepts'], 'one', j['prove_seconds_one_context'], 'group', j['prove_seconds_group'], j['group_breakdown_seconds_max_over_contexts'], [ (p['columns_transformed_here'], p['exchange_bytes_received']>>20) for p in j['per_context']])"
done
