#!/bin/bash
# per-stream scratch arenas (this build) x early random-polynomial commitment x grouping, on the bench circuits
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  L=$1; shift
  (cd "$R" && env "$@" REPS=8 timeout 300 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$L', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f rnd %.4f h %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['random_poly'], b['h_split_commit']), j['proof_sha256'])"
}
M="CIRCUIT=mlp K=20"
run "mlp20 early" $M
run "mlp20 off" $M EZKL_PROVER_NO_EARLY_RANDOM=1
run "mlp20 early BIG=1" $M EZKL_MSM_GROUP_BIG=1
run "mlp20 off BIG=1" $M EZKL_PROVER_NO_EARLY_RANDOM=1 EZKL_MSM_GROUP_BIG=1
run "mlp20 early merged" $M EZKL_PROVER_MERGED_COMMITS=1
run "einsum20 early" CIRCUIT=einsum K=20
run "einsum20 off" CIRCUIT=einsum K=20 EZKL_PROVER_NO_EARLY_RANDOM=1
run "mlp17 early" CIRCUIT=mlp K=17
run "mlp17 off" CIRCUIT=mlp K=17 EZKL_PROVER_NO_EARLY_RANDOM=1
run "conv17 early" CIRCUIT=conv K=17
run "conv17 off" CIRCUIT=conv K=17 EZKL_PROVER_NO_EARLY_RANDOM=1
