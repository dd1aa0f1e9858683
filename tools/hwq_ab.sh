#!/bin/bash
# GPU_MAX_HW_QUEUES (HIP hardware queues the runtime multiplexes the library's ~11 streams onto) x fused general-scalar groups, k = 20 MLP proof
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  L=$1; shift
  (cd "$R" && env "$@" CIRCUIT=mlp K=20 REPS=8 timeout 300 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$L', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f h %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['h_split_commit']), j['proof_sha256'])"
}
for Q in 8 12 16 24; do run "HWQ=$Q default groups" GPU_MAX_HW_QUEUES=$Q; done
for Q in 12 16; do run "HWQ=$Q BIG=4" GPU_MAX_HW_QUEUES=$Q EZKL_MSM_GROUP_BIG=4; done
run "HWQ=4 BIG=4" GPU_MAX_HW_QUEUES=4 EZKL_MSM_GROUP_BIG=4
