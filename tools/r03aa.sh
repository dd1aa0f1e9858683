#!/bin/bash
# key loader as a pread -> pinned -> HBM pipeline, indicator columns filled on the device: tests + the cold one-shot of the k = 20 MLP
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_native_prover.py tests/test_execute.py -m gpu -x -q > "$O/r03aa_pytest.log" 2>&1; tail -3 "$O/r03aa_pytest.log"
CIRCUIT=mlp K=20 REPS=3 timeout 900 python tools/prove_bench.py --pinned --cold > "$O/r03aa_cold_mlp20.log" 2>&1
grep '^{' "$O/r03aa_cold_mlp20.log" | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['prove_seconds_gpu'], j['keygen_seconds_gpu'], j['proof_sha256']); print(json.dumps(j.get('cold'))[:1500])"
