#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
(cd "$R" && EZKL_MSM_DEBUG=1 CIRCUIT=mlp K=20 REPS=1 timeout 300 python tools/prove_bench.py --pinned) 2>&1 | grep "msm batch" | tail -12
