#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp
(time CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=3 timeout 1500 python "$R/tools/prove_bench.py" --pinned) > "$O/r03o_k22.log" 2>&1
tail -5 "$O/r03o_k22.log" | cut -c1-2500
