#!/bin/bash
# z committed while the lookup sums are computed (msm_batch push_many): MSM batch tests, prover tests, proof time + bytes (merged | split)
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "batch or upload" > "$O/r03ae_pytest_msm.log" 2>&1; tail -3 "$O/r03ae_pytest_msm.log"
for mode in merged split; do
  [ $mode = split ] && export EZKL_PROVER_SPLIT_COMMITS=1
  CIRCUIT=mlp K=20 REPS=8 timeout 600 python tools/prove_bench.py --pinned > "$O/r03ae_mlp20_$mode.log" 2>&1
  grep '^{' "$O/r03ae_mlp20_$mode.log" | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$mode', j['prove_seconds_gpu_runs'], j['proof_sha256'], j['prove_breakdown_seconds'])"
done
unset EZKL_PROVER_SPLIT_COMMITS
CIRCUIT=mlp K=17 REPS=8 timeout 600 python tools/prove_bench.py --pinned > "$O/r03ae_mlp17.log" 2>&1
grep '^{' "$O/r03ae_mlp17.log" | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('k17', j['prove_seconds_gpu_runs'], j['proof_sha256'])"
timeout 900 python -m pytest tests/test_native_prover.py tests/test_ezkl_circuit.py -m gpu -x -q > "$O/r03ae_pytest.log" 2>&1; tail -3 "$O/r03ae_pytest.log"
