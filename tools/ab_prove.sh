#!/bin/bash
# A/B of the end-to-end prove on ONE box: a previous build (its two .so files copied into ab_old/, git-ignored but shipped by gpurun)
# against the in-tree build, alternating; usage (on the GPU box): bash tools/ab_prove.sh
for i in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then export EZKL_HIP_LIB=$PWD/ab_old/libezkl_hip.so EZKL_PROVER_LIB=$PWD/ab_old/libezkl_prover.so; else unset EZKL_HIP_LIB EZKL_PROVER_LIB; fi
    K=${K:-20} BLOCKS=4 python tools/prove_bench.py --native --pinned 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); nv=j['native_prover']
print('$v', 'py', j['prove_seconds_gpu'], j['prove_breakdown_seconds'], 'native', nv['prove_seconds_library_rng'], nv['breakdown_seconds_library_rng'])"
  done
done
