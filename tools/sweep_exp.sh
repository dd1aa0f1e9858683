# quotient-sweep JIT code-generation experiment (NOTEBOOK.md §4.3): EZKL_EVALH_{NO_SCHEDULE,WAVES,BARRIER,R29} on the MLP circuit
#   CFGS='A=1,B=2 C=3' bash tools/sweep_exp.sh   (configurations separated by spaces, assignments inside one by commas)
export EZKL_HIP_CACHE_DIR=off
for cfg in ${CFGS:-X=1}; do
  echo "== $cfg"
  env $(echo $cfg | tr ',' ' ') CIRCUIT=${CIRCUIT:-mlp} K=${K:-18} REPS=3 python tools/prove_bench.py --native --pinned 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['prove_seconds_gpu'], j['prove_breakdown_seconds']['quotient_sweep'], j['proof_sha256'])"
done
