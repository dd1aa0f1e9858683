mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_native_prover.py tests/test_gpu_msm.py -m gpu -q --timeout 500 > gpurun_out/r05t_pytest_sub.log 2>&1; echo "sub rc=$?" ); tail -4 gpurun_out/r05t_pytest_sub.log
echo "=== integer rep"; CIRCUIT=mlp K=20 REPS=6 timeout 400 python tools/prove_bench.py --pinned --integer-rep 2>/dev/null | tail -1 > gpurun_out/r05t_mlp_k20_intrep.log
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05t_mlp_k20_intrep.log').read())
print(j['prove_seconds_gpu_runs'], j['prove_breakdown_seconds']); print(j.get('integer_rep_advice'))
PY
echo "=== skew"; bash tools/run.sh r05t ab:skew 2>&1 | tail -12 | cut -c1-330
