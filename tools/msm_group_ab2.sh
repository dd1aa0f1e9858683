#!/bin/bash
# EZKL_MSM_GROUP_BIG (general-scalar commit batches fused in groups) across the bench circuits: default (no fusing at 2^20 and above) vs 4
R=$(cd "$(dirname "$0")/.." && pwd)
run() {   # label, env...
  L=$1; shift
  (cd "$R" && env "$@" REPS=6 timeout 600 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$L', sorted(j['prove_seconds_gpu_runs'])[:3], ' '.join('%s %.4f' % (a[:6], v) for a, v in b.items() if a != 'total'), j['proof_sha256'])"
}
for BIG in 0 4 3; do
  X="EZKL_MSM_GROUP_BIG=$BIG"
  run "mlp20 BIG=$BIG" $X CIRCUIT=mlp K=20
  run "einsum20 BIG=$BIG" $X CIRCUIT=einsum K=20
  run "mlp17 BIG=$BIG" $X CIRCUIT=mlp K=17
  run "conv17 BIG=$BIG" $X CIRCUIT=conv K=17
  run "mlp22 BIG=$BIG" $X CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25
done
