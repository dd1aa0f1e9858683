#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp
CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=3 timeout 1500 python "$R/tools/prove_bench.py" --pinned > "$O/r03s_k22.log" 2>&1
tail -1 "$O/r03s_k22.log" | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['prove_seconds_gpu_runs'], j['prove_breakdown_seconds'], j['hbm_in_use_gib_after_prove'], j['first_prove_seconds_gpu'], j['keygen_seconds_gpu'])"
