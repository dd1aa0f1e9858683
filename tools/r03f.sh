#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd "$R"
python - > "$O/r03f_ubench.log" 2>&1 <<'PY'
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
print("check one-chain mismatches:", B.ubench("modmul29_check"), " two-chain mismatches:", B.ubench("modmul29i_check"))
for occ in ("_o1", "_o2", "_o3", "_o4", "_o6", ""):
    a, b = B.ubench("modmul29" + occ), B.ubench("modmul29i" + occ)
    print("waves/SIMD %-4s one-chain %.3e  two-chain %.3e  ratio %.3f" % (occ or "max", a, b, b / a))
PY
cat "$O/r03f_ubench.log"
timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -x -q > "$O/r03f_pytest.log" 2>&1; tail -2 "$O/r03f_pytest.log"
cd /tmp && export TMPDIR=/tmp
for V in new old; do
  if [ $V = old ]; then export EZKL_HIP_LIB=$R/ab_old/libezkl_hip.so EZKL_PROVER_LIB=$R/ab_old/libezkl_prover.so; fi
  (cd $R && python bench.py --no-cpu-baseline --with-batch) > "$O/r03f_bench_$V.log" 2>&1
  python - "$O/r03f_bench_$V.log" $V <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "pts/s %.4g ms/step %.4f acc_ms %.4f msm_dev_ms %.4f batch4 %.4f" % (j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["extra"]["msm_device_ms"], j["extra"].get("msm_batch4_ms_per_msm", 0)))
PY
  CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03f_prove_$V.log" 2>&1
  echo "$V $(tail -1 $O/r03f_prove_$V.log | grep -o '"prove_seconds_gpu_runs": [^]]*]') $(tail -1 $O/r03f_prove_$V.log | grep -o '"prove_breakdown_seconds": {[^}]*}') $(tail -1 $O/r03f_prove_$V.log | grep -o '"proof_sha256": "[0-9a-f]*"')"
done
unset EZKL_HIP_LIB EZKL_PROVER_LIB
timeout 600 rocprofv3 --kernel-trace -d "$O/r03f_msmcols" -- python "$R/tools/msm_columns_profile.py" run > "$O/r03f_msmcols.log" 2>&1
DB=$(find "$O/r03f_msmcols" -name '*.db' | head -1)
python "$R/tools/msm_columns_profile.py" reduce "$DB" "$O/r03f_msmcols.log" > "$O/r03f_msmcols_table.txt" 2>&1
rm -rf "$O/r03f_msmcols"; cat "$O/r03f_msmcols_table.txt"
