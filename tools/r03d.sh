#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() { # label, env...
  L=$1; shift
  env "$@" CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03d_$L.log" 2>&1
  echo "$L $(tail -1 $O/r03d_$L.log | grep -o '"prove_seconds_gpu_runs": [^]]*]') $(tail -1 $O/r03d_$L.log | grep -o '"prove_breakdown_seconds": {[^}]*}')"
}
run base X=1
run group4 EZKL_MSM_GROUP=4
run group2 EZKL_MSM_GROUP=2
run group4_q16 EZKL_MSM_GROUP=4 GPU_MAX_HW_QUEUES=16
run slots1 EZKL_MSM_SLOTS=1
run slots3 EZKL_MSM_SLOTS=3
run slots12 EZKL_MSM_SLOTS=12 GPU_MAX_HW_QUEUES=16
