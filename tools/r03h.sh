#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests/test_multi_owner.py tests/test_plonk.py -m gpu -x -q -k "owner or replicated or bench_contract" > "$O/r03h_pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/r03h_pytest.log"
tail -5 "$O/r03h_pytest.log"
(time python bench.py) > "$O/r03h_bench.log" 2>&1
tail -4 "$O/r03h_bench.log" | cut -c1-3000
bash tools/pmc_prove.sh r03h > "$O/r03h_pmc.log" 2>&1; tail -20 "$O/r03h_pmc.log"
