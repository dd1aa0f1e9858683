#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_ntt.py tests/test_native_prover.py tests/test_gpu_misc.py -m gpu -x -q > "$O/r03j_pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/r03j_pytest.log"
tail -8 "$O/r03j_pytest.log"
python tools/ntt_ab.py 2>&1 | grep "k=" | sed 's/^/maxr10 /'
EZKL_NTT_MAXR=8 python tools/ntt_ab.py 2>&1 | grep "k=" | sed 's/^/maxr8  /'
cd /tmp && export TMPDIR=/tmp
for V in 10 8; do
EZKL_NTT_MAXR=$V CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03j_prove$V.log" 2>&1
echo "maxr$V $(tail -1 $O/r03j_prove$V.log | grep -o '"prove_seconds_gpu_runs": [^]]*]') $(tail -1 $O/r03j_prove$V.log | grep -o '"proof_sha256": "[0-9a-f]*"')"
done
