#!/bin/bash
# code-generation knobs of the quotient sweep on the k = 20 MLP key: sweep kernel time per coset (HIP events) and whole-proof time
R=$(cd "$(dirname "$0")/.." && pwd)
run() {
  L=$1; shift
  (cd "$R" && env "$@" CIRCUIT=mlp K=20 REPS=6 timeout 300 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$L', sorted(j['prove_seconds_gpu_runs'])[:3], 'sweep ms/coset', j['sweep_kernel']['avg_launch_ms'], 'h stage', j['prove_breakdown_seconds']['h_split_commit'], j['proof_sha256'])"
}
run "defaults" A=1
for N in 2 3 4 8; do run "BARRIER_EVERY=$N" EZKL_EVALH_BARRIER_EVERY=$N; done
for W in 3 5 6; do run "WAVES=$W" EZKL_EVALH_WAVES=$W; done
run "XCD=1" EZKL_EVALH_XCD=1
run "R29=1 (inlined product)" EZKL_EVALH_R29=1
