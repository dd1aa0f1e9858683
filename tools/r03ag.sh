#!/bin/bash
# the in-process prover group at k = 20 after the round's host-side changes (staged uploads, asynchronous sweeps): 2 and 4 contexts on one
# device, and two gloo ranks sharing the device in owner mode -- every one must emit the one-GPU bytes
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp
CONTEXTS=0,0 CIRCUIT=mlp K=20 REPS=2 timeout 600 python "$R/tools/prove_group.py" --pinned > "$O/r03ag_group2.log" 2>&1; tail -1 "$O/r03ag_group2.log" | cut -c1-700
CONTEXTS=0,0,0,0 CIRCUIT=mlp K=20 REPS=2 timeout 600 python "$R/tools/prove_group.py" --pinned > "$O/r03ag_group4.log" 2>&1; tail -1 "$O/r03ag_group4.log" | cut -c1-700
cd "$R"
CIRCUIT=mlp K=20 REPS=2 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29621 tools/prove_multi.py --gloo --share-device --pinned > "$O/r03ag_multi2.log" 2>&1; grep '^{' "$O/r03ag_multi2.log" | tail -1 | cut -c1-900
