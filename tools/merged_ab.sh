#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  L=$1; shift
  (cd "$R" && env "$@" CIRCUIT=mlp K=20 REPS=8 timeout 300 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$L', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f h %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['h_split_commit']), j['proof_sha256'])"
}
run "defaults" A=1
run "MERGED_COMMITS=1" EZKL_PROVER_MERGED_COMMITS=1
run "MERGED_COMMITS=1 HWQ=12" EZKL_PROVER_MERGED_COMMITS=1 GPU_MAX_HW_QUEUES=12
run "SYNC_CALLS=1" EZKL_PROVER_SYNC_CALLS=1
