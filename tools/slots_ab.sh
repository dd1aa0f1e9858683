#!/bin/bash
# how many streams exist at all: EZKL_MSM_SLOTS (batch slots) x grouping, with the early random commitment and per-stream arenas
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  L=$1; shift
  (cd "$R" && env "$@" timeout 600 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$L', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f rnd %.4f h %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['random_poly'], b['h_split_commit']), 'keygen', j['keygen_seconds_gpu'], j['proof_sha256'])"
}
for S in 2 3 4; do
  for BIG in 1 2 4; do
    run "mlp20 SLOTS=$S BIG=$BIG" EZKL_MSM_SLOTS=$S EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=mlp K=20 REPS=10
  done
done
run "mlp20 SLOTS=3 BIG=2 SMALL=4" EZKL_MSM_SLOTS=3 EZKL_MSM_GROUP_BIG=2 EZKL_MSM_GROUP_SMALL=4 CIRCUIT=mlp K=20 REPS=10
run "mlp20 SLOTS=2 BIG=1 HWQ=12" EZKL_MSM_SLOTS=2 EZKL_MSM_GROUP_BIG=1 GPU_MAX_HW_QUEUES=12 CIRCUIT=mlp K=20 REPS=10
run "einsum20 SLOTS=3 BIG=2" EZKL_MSM_SLOTS=3 EZKL_MSM_GROUP_BIG=2 CIRCUIT=einsum K=20 REPS=10
run "einsum20 SLOTS=2 BIG=4" EZKL_MSM_SLOTS=2 EZKL_MSM_GROUP_BIG=4 CIRCUIT=einsum K=20 REPS=10
run "mlp17 SLOTS=3" EZKL_MSM_SLOTS=3 CIRCUIT=mlp K=17 REPS=10
run "mlp17 SLOTS=2" EZKL_MSM_SLOTS=2 CIRCUIT=mlp K=17 REPS=10
