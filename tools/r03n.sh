#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp
run() { L=$1; shift
  env "$@" EZKL_HIP_CACHE_DIR=off CIRCUIT=mlp K=20 REPS=4 timeout 900 python "$R/tools/prove_bench.py" --pinned > "$O/r03n_$L.log" 2>&1
  echo "$L $(tail -1 $O/r03n_$L.log | grep -o '"prove_seconds_gpu": [0-9.]*') $(tail -1 $O/r03n_$L.log | grep -o '"quotient_sweep": [0-9.]*') $(tail -1 $O/r03n_$L.log | grep -o '"sweep_kernel": {[^}]*}' | grep -o '"avg_launch_ms": [0-9.]*') $(tail -1 $O/r03n_$L.log | grep -o '"proof_sha256": "[0-9a-f]*"')"
}
run every1 EZKL_EVALH_BARRIER_EVERY=1
run every2 EZKL_EVALH_BARRIER_EVERY=2
run every3 EZKL_EVALH_BARRIER_EVERY=3
run every4 EZKL_EVALH_BARRIER_EVERY=4
run every2_w3 EZKL_EVALH_BARRIER_EVERY=2 EZKL_EVALH_WAVES=3
run every8_w2 EZKL_EVALH_BARRIER_EVERY=8 EZKL_EVALH_WAVES=2
run w5 EZKL_EVALH_WAVES=5
