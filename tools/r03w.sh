#!/bin/bash
# after the persistent side stream / pinned staging / device-side lookup counter: proof time + bytes, host trace, the prover tests
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
CIRCUIT=mlp K=20 REPS=8 timeout 600 python tools/prove_bench.py --pinned > "$O/r03w_mlp20.log" 2>&1; tail -3 "$O/r03w_mlp20.log" | cut -c1-600
CIRCUIT=einsum K=20 REPS=8 timeout 600 python tools/prove_bench.py --pinned > "$O/r03w_einsum20.log" 2>&1; tail -2 "$O/r03w_einsum20.log" | cut -c1-400
timeout 1500 python -m pytest tests/test_native_prover.py tests/test_gpu_misc.py tests/test_cpp_mirror.py tests/test_gpu_evalh.py tests/test_group.py tests/test_multi_owner.py -m gpu -x -q > "$O/r03w_pytest.log" 2>&1; tail -4 "$O/r03w_pytest.log"
cd /tmp && export TMPDIR=/tmp
CIRCUIT=mlp K=20 REPS=3 timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d "$O/r03w_prove" -- python "$R/tools/prove_bench.py" --pinned > "$O/r03w_prove.log" 2>&1
DB=$(find "$O/r03w_prove" -name '*.db' | head -1)
python "$R/tools/hosttrace.py" "$DB" 95 120 > "$O/r03w_prove_hosttrace.txt" 2>&1
python "$R/tools/timeline.py" "$DB" 95 > "$O/r03w_prove_timeline.txt" 2>&1
rm -rf "$O/r03w_prove"
head -8 "$O/r03w_prove_timeline.txt"; grep -A14 "^entry point" "$O/r03w_prove_hosttrace.txt"
