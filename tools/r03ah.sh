#!/bin/bash
# key forms computed under the readers: the key-file tests and the cold one-shot again
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_native_prover.py tests/test_execute.py -m gpu -x -q -k "key or file or execute or setup or prove" > "$O/r03ah_pytest.log" 2>&1; tail -2 "$O/r03ah_pytest.log"
CIRCUIT=mlp K=20 REPS=2 timeout 900 python tools/prove_bench.py --pinned --cold > "$O/r03ah_cold_mlp20.log" 2>&1
grep '^{' "$O/r03ah_cold_mlp20.log" | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); c = j['cold']; print(j['prove_seconds_gpu'], j['proof_sha256'], c['cold_seconds'], c['stages'], c['same_proof_as_warm'], c['first_ever_cold_seconds'])"
