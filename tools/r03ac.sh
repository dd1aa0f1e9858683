#!/bin/bash
# k = 22, 30 advice columns, every cell assigned: keygen stage times + proof after the host-side fixes
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
EZKL_PROVER_KEYGEN_TIMING=1 CIRCUIT=mlp K=22 MLP_BLOCKS=5 REPS=3 timeout 900 python tools/prove_bench.py --pinned > "$O/r03ac_k22_full.log" 2>&1
grep "keygen" "$O/r03ac_k22_full.log" | head -12
grep '^{' "$O/r03ac_k22_full.log" | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['prove_seconds_gpu_runs'], j['keygen_seconds_gpu'], j['first_prove_seconds_gpu'], j['proof_sha256'], j['prove_breakdown_seconds'], j['sweep_kernel'])"
