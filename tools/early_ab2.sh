#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  L=$1; shift
  (cd "$R" && env "$@" timeout 600 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$L', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f rnd %.4f h %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['random_poly'], b['h_split_commit']), j['proof_sha256'])"
}
for E in early off; do
  if [ $E = off ]; then X="EZKL_PROVER_NO_EARLY_RANDOM=1"; else X="A=1"; fi
  for BIG in 1 2 4; do
    run "mlp20 $E BIG=$BIG" $X EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=mlp K=20 REPS=10
    run "einsum20 $E BIG=$BIG" $X EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=einsum K=20 REPS=10
  done
done
for BIG in 1 4; do
  run "mlp22 early BIG=$BIG" A=1 EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=3
done
run "mlp20 early BIG=1 SMALL=4" EZKL_MSM_GROUP_BIG=1 EZKL_MSM_GROUP_SMALL=4 CIRCUIT=mlp K=20 REPS=10
run "mlp20 early BIG=1 SMALL=8" EZKL_MSM_GROUP_BIG=1 EZKL_MSM_GROUP_SMALL=8 CIRCUIT=mlp K=20 REPS=10
run "mlp20 early BIG=1 HWQ=12" EZKL_MSM_GROUP_BIG=1 GPU_MAX_HW_QUEUES=12 CIRCUIT=mlp K=20 REPS=10
run "mlp20 early BIG=1 merged" EZKL_MSM_GROUP_BIG=1 EZKL_PROVER_MERGED_COMMITS=1 CIRCUIT=mlp K=20 REPS=10
