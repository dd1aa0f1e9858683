#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_contexts_cpu.py tests/test_execute.py -x -q > "$O/r03l_pytest.log" 2>&1; echo "rc=$?" >> "$O/r03l_pytest.log"; tail -12 "$O/r03l_pytest.log"
cd /tmp
CONTEXTS=0,0 CIRCUIT=mlp K=20 REPS=3 timeout 900 python "$R/tools/prove_group.py" --pinned > "$O/r03l_group2.log" 2>&1; tail -1 "$O/r03l_group2.log" | cut -c1-1800
CONTEXTS=0,0,0,0 CIRCUIT=mlp K=17 REPS=3 timeout 900 python "$R/tools/prove_group.py" --pinned > "$O/r03l_group4.log" 2>&1; tail -1 "$O/r03l_group4.log" | cut -c1-1200
CIRCUIT=mlp K=20 REPS=3 EZKL_COLD_DIR=/tmp timeout 1200 python "$R/tools/prove_bench.py" --pinned --cold > "$O/r03l_cold_mlp20.log" 2>&1
tail -1 "$O/r03l_cold_mlp20.log" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'warm': j['prove_seconds_gpu'], 'cold': j.get('cold')}))" | cut -c1-1500
