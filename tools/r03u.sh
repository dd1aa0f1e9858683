#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp
(time CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=2 timeout 2400 python "$R/tools/prove_bench.py" --pinned --cpu) > "$O/r03u_k22_cpu.log" 2>&1
tail -5 "$O/r03u_k22_cpu.log" | cut -c1-3000
