#!/bin/bash
# A/B of the fused-group size of a commit phase (EZKL_MSM_GROUP) on the k = 20 MLP proof: does ONE chain per phase beat groups of four?
R=$(cd "$(dirname "$0")/.." && pwd)
for G in default 2 6 8 12; do
  if [ $G = default ]; then unset EZKL_MSM_GROUP; else export EZKL_MSM_GROUP=$G; fi
  (cd "$R" && CIRCUIT=mlp K=20 REPS=6 timeout 300 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('EZKL_MSM_GROUP=$G', j['prove_seconds_gpu_runs'], j['prove_breakdown_seconds'], j['proof_sha256'])"
done
