#!/bin/bash
# upload_end on events; the FULL-fill k = 22 30-column proof (every advice cell assigned); bench contract at world 2
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
CIRCUIT=mlp K=20 REPS=8 timeout 600 python tools/prove_bench.py --pinned > "$O/r03y_mlp20.log" 2>&1
timeout 900 python -m pytest tests/test_plonk.py -m gpu -q -k bench_contract > "$O/r03y_pytest.log" 2>&1; tail -2 "$O/r03y_pytest.log"
CIRCUIT=mlp K=22 MLP_BLOCKS=5 REPS=3 timeout 1200 python tools/prove_bench.py --pinned > "$O/r03y_k22_full.log" 2>&1
for f in r03y_mlp20 r03y_k22_full; do tail -1 "$O/$f.log" | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['circuit'].get('cells_used'), j['circuit'].get('advice_columns'), j['prove_seconds_gpu_runs'], j['prove_breakdown_seconds'], j['hbm_in_use_gib_after_prove'], j['first_prove_seconds_gpu'], j['keygen_seconds_gpu'], j['proof_sha256'])"; done
