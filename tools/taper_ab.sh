#!/bin/bash
# upload phases in tapered groups (6 + 3 + 2 + 1) against equal groups (6 + 6)
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  L=$1; shift
  (cd "$R" && env "$@" timeout 600 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$L', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f h %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['h_split_commit']), j['proof_sha256'])"
}
for T in taper equal; do
  if [ $T = equal ]; then X="EZKL_MSM_NO_TAPER=1"; else X="A=1"; fi
  run "mlp20 $T" $X CIRCUIT=mlp K=20 REPS=10
  run "einsum20 $T" $X CIRCUIT=einsum K=20 REPS=10
  run "mlp17 $T" $X CIRCUIT=mlp K=17 REPS=10
  run "conv17 $T" $X CIRCUIT=conv K=17 REPS=10
  run "mlp22 $T" $X CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=3
done
run "mlp20 taper SMALL=4" EZKL_MSM_GROUP_SMALL=4 CIRCUIT=mlp K=20 REPS=10
run "mlp20 taper SMALL=8" EZKL_MSM_GROUP_SMALL=8 CIRCUIT=mlp K=20 REPS=10
