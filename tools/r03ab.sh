#!/bin/bash
# keygen stage times (k = 20 MLP read from the layout cache; k = 22 30-column if the cache travelled)
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
EZKL_PROVER_KEYGEN_TIMING=1 CIRCUIT=mlp K=20 REPS=2 timeout 600 python tools/prove_bench.py --pinned > "$O/r03ab_mlp20.log" 2>&1
grep "keygen" "$O/r03ab_mlp20.log" | head -12
grep '^{' "$O/r03ab_mlp20.log" | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['prove_seconds_gpu'], j['keygen_seconds_gpu'], j['proof_sha256'], j['sweep_kernel'])"
