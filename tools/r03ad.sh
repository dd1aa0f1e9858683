#!/bin/bash
# sigma columns gathered on the device: kernel test, key parity tests (native key = Python key = the reference's pk.key), keygen stage times
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_misc.py tests/test_native_prover.py tests/test_execute.py tests/test_ezkl_circuit.py tests/test_group.py -m gpu -x -q > "$O/r03ad_pytest.log" 2>&1; tail -3 "$O/r03ad_pytest.log"
EZKL_PROVER_KEYGEN_TIMING=1 CIRCUIT=mlp K=20 REPS=2 timeout 600 python tools/prove_bench.py --pinned > "$O/r03ad_mlp20.log" 2>&1
grep "keygen" "$O/r03ad_mlp20.log" | head -12
grep '^{' "$O/r03ad_mlp20.log" | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['prove_seconds_gpu'], j['keygen_seconds_gpu'], j['proof_sha256'])"
