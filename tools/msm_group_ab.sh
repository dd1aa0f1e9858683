#!/bin/bash
# A/B of the fused-group size of the witness-shaped commit phases (EZKL_MSM_GROUP_SMALL: advice columns, multiplicities) on the k = 20 MLP
# proof, and of EZKL_MSM_GROUP (every phase) -- does one chain per phase beat groups of four?
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  (cd "$R" && CIRCUIT=mlp K=20 REPS=8 timeout 300 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$1', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f h %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['h_split_commit']), j['proof_sha256'])"
}
for ONLY in 5 8 4 1; do export EZKL_MSM_GROUP_BIG=4 EZKL_MSM_GROUP_BIG_ONLY=$ONLY; run "BIG=4 only for batches of $ONLY"; done
unset EZKL_MSM_GROUP_BIG EZKL_MSM_GROUP_BIG_ONLY
run "defaults"
