#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
EZKL_COLD_AB=EZKL_PROVER_PK_FORMS_AFTER=1 EZKL_COLD_AB_REPS=4 CIRCUIT=mlp K=20 REPS=2 timeout 900 python tools/prove_bench.py --pinned --cold > "$O/r03ai_cold_ab.log" 2>&1
grep '^{' "$O/r03ai_cold_ab.log" | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); c = j['cold']; print(c['cold_seconds'], c['stages'])
for k, v in c['ab'].items():
    print(k); [print('   ', x) for x in v]"
