#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_ntt.py tests/test_native_prover.py -m gpu -x -q > "$O/r03e_pytest.log" 2>&1; echo "pytest rc=$?" >> "$O/r03e_pytest.log"
tail -4 "$O/r03e_pytest.log"
cd /tmp && export TMPDIR=/tmp
run() { L=$1; shift
  env "$@" CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03e_$L.log" 2>&1
  echo "$L $(tail -1 $O/r03e_$L.log | grep -o '"prove_seconds_gpu_runs": [^]]*]') $(tail -1 $O/r03e_$L.log | grep -o '"prove_breakdown_seconds": {[^}]*}') $(tail -1 $O/r03e_$L.log | grep -o '"proof_sha256": "[0-9a-f]*"')"
}
run base X=1
run lmin16 EZKL_MSM_LMIN=16
run lmin32 EZKL_MSM_LMIN=32
run lmin32_span4 EZKL_MSM_LMIN=32 EZKL_MSM_SPAN=4
run span4 EZKL_MSM_SPAN=4
run q16 GPU_MAX_HW_QUEUES=16
for LM in 8 32; do
EZKL_MSM_LMIN=$LM timeout 600 rocprofv3 --kernel-trace -d "$O/r03e_msmcols$LM" -- python "$R/tools/msm_columns_profile.py" run > "$O/r03e_msmcols$LM.log" 2>&1
DB=$(find "$O/r03e_msmcols$LM" -name '*.db' | head -1)
python "$R/tools/msm_columns_profile.py" reduce "$DB" "$O/r03e_msmcols$LM.log" > "$O/r03e_msmcols${LM}_table.txt" 2>&1
rm -rf "$O/r03e_msmcols$LM"; cat "$O/r03e_msmcols${LM}_table.txt"
done
