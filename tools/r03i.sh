#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() { L=$1; shift
  env "$@" CIRCUIT=mlp K=20 REPS=5 timeout 900 python "$R/tools/prove_bench.py" --pinned > "$O/r03i_$L.log" 2>&1
  echo "$L $(tail -1 $O/r03i_$L.log | grep -o '"prove_seconds_gpu_runs": [^]]*]') $(tail -1 $O/r03i_$L.log | grep -o '"quotient_sweep": [0-9.]*') $(tail -1 $O/r03i_$L.log | grep -o '"keygen_seconds_gpu": [0-9.]*') $(tail -1 $O/r03i_$L.log | grep -o '"first_prove_seconds_gpu": [0-9.]*') $(tail -1 $O/r03i_$L.log | grep -o '"proof_sha256": "[0-9a-f]*"')"
}
# keygen_seconds includes hiprtc of the sweep kernels when the on-disk cache is off
run all1 EZKL_HIP_CACHE_DIR=off EZKL_PROVER_SWEEP_TERMS=0
run t32 EZKL_HIP_CACHE_DIR=off EZKL_PROVER_SWEEP_TERMS=32
run t16 EZKL_HIP_CACHE_DIR=off EZKL_PROVER_SWEEP_TERMS=16
run t8 EZKL_HIP_CACHE_DIR=off EZKL_PROVER_SWEEP_TERMS=8
run t4 EZKL_HIP_CACHE_DIR=off EZKL_PROVER_SWEEP_TERMS=4
