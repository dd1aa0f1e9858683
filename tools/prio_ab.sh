#!/bin/bash
# stream priorities: library stream (helper chains on the critical path) / side stream (NTT forms, needed late) / MSM slot streams
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  L=$1; shift
  (cd "$R" && env "$@" CIRCUIT=mlp K=20 REPS=8 timeout 300 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']; print('$L', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f h %.4f ntt %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['h_split_commit'], b['intt_and_coset_ntt']), j['proof_sha256'])"
}
run "defaults" A=1
run "LIB high" EZKL_HIP_PRIO_LIB=-1
run "AUX low" EZKL_HIP_PRIO_AUX=1
run "LIB high AUX low" EZKL_HIP_PRIO_LIB=-1 EZKL_HIP_PRIO_AUX=1
run "LIB high MSM high AUX low" EZKL_HIP_PRIO_LIB=-1 EZKL_HIP_PRIO_MSM=-1 EZKL_HIP_PRIO_AUX=1
run "MSM high" EZKL_HIP_PRIO_MSM=-1
run "AUX high" EZKL_HIP_PRIO_AUX=-1
