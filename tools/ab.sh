#!/bin/bash
# (round 6: EZKL_MSM_COOP / ZEROCOPY / ORDER_ALWAYS / NO_TAPER / FIXUP_TREE and the compile-time EZKL_MSM_* / EZKL_NTT_* variants were removed from the
#  sources after their measurements -- the experiments below that name them are history; their logs are in profiles/.)
# tools/ab.sh <experiment> -- the A/B experiments of round 4 behind NOTEBOOK.md §4.1.3 / §4.3 (one script, one `run` helper; each line of
# output is one setting: best proof times of REPS proofs, the stage times of the best one, the proof hash -- identical everywhere).
#   bash tools/run.sh <tag> ab:<experiment>     on the GPU box; the log lands in gpurun_out/<tag>_ab_<experiment>.log
# experiments: group (fused-group sizes, k = 20 MLP) | circuits (general-scalar groups across the bench circuits) | hwq (hardware queues)
#   | prio (stream priorities) | evalh (sweep code generation) | merged (z committed under the lookup sums) | early (random polynomial
#   committed under the witness upload x priorities) | matrix (early x groups, k = 20 / 22) | slots (batch slots x groups) | taper (upload
#   phases in tapered groups) | benchvar (bench.py twice with 4 and 6 slots: box-to-box and run-to-run variation) | msmdebug (the batches
#   of one proof as the library sees them) | sumscatter (SHPLONK commitments reduce-scattered across contexts) | ntt29r (pass radices under the radix-2^29 NTT pass; the A/B against the radix-2^32 pass it replaced is profiles/r04ai_ab_ntt29.log, run at commit 'NTT: radix-2^29 decimation-in-time pass' where both existed)
R=$(cd "$(dirname "$0")/.." && pwd)
run() {   # label, then VAR=value ... (CIRCUIT / K / REPS included)
  L=$1; shift
  (cd "$R" && env "$@" timeout 600 python tools/prove_bench.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); b = j['prove_breakdown_seconds']
print('$L', sorted(j['prove_seconds_gpu_runs'])[:4], 'advice %.4f m %.4f z %.4f phi %.4f rnd %.4f h %.4f' % (b['advice_commit'], b['lookup_m'], b['permutation_z'], b['lookup_phi'], b['random_poly'], b['h_split_commit']),
      'sweep ms/coset %.4f' % j['sweep_kernel']['avg_launch_ms'], 'keygen', j['keygen_seconds_gpu'], j['proof_sha256'])"
}
M="CIRCUIT=mlp K=20 REPS=8"
case "$1" in
  group)
    run "defaults" $M
    for G in 4 5 6 7 8; do run "GROUP_SMALL=$G" $M EZKL_MSM_GROUP_SMALL=$G; done
    for G in 1 2 3 4 6; do run "GROUP_BIG=$G" $M EZKL_MSM_GROUP_BIG=$G; done
    for O in 5 8 4 1; do run "GROUP_BIG=4 only for batches of $O" $M EZKL_MSM_GROUP_BIG=4 EZKL_MSM_GROUP_BIG_ONLY=$O; done ;;
  circuits)
    for BIG in 1 4 3; do
      run "mlp20 BIG=$BIG" EZKL_MSM_GROUP_BIG=$BIG $M; run "einsum20 BIG=$BIG" EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=einsum K=20 REPS=6
      run "mlp17 BIG=$BIG" EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=mlp K=17 REPS=6; run "conv17 BIG=$BIG" EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=conv K=17 REPS=6
      run "mlp22 BIG=$BIG" EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=3
    done ;;
  hwq)
    for Q in 4 8 12 16 24; do run "HWQ=$Q" $M GPU_MAX_HW_QUEUES=$Q; done
    for Q in 12 16; do run "HWQ=$Q GROUP_BIG=1" $M GPU_MAX_HW_QUEUES=$Q EZKL_MSM_GROUP_BIG=1; done ;;
  prio)
    run "defaults" $M; run "LIB high" $M EZKL_HIP_PRIO_LIB=-1; run "AUX low" $M EZKL_HIP_PRIO_AUX=1
    run "LIB high AUX low" $M EZKL_HIP_PRIO_LIB=-1 EZKL_HIP_PRIO_AUX=1; run "LIB high MSM high AUX low" $M EZKL_HIP_PRIO_LIB=-1 EZKL_HIP_PRIO_MSM=-1 EZKL_HIP_PRIO_AUX=1
    run "MSM high" $M EZKL_HIP_PRIO_MSM=-1; run "AUX high" $M EZKL_HIP_PRIO_AUX=-1 ;;
  evalh)
    run "defaults" $M
    for N in 1 2 3 4 8; do run "BARRIER_EVERY=$N" $M EZKL_EVALH_BARRIER_EVERY=$N; done
    for W in 3 5 6; do run "WAVES=$W" $M EZKL_EVALH_WAVES=$W; done
    run "XCD=1" $M EZKL_EVALH_XCD=1; run "R29=1 (inlined product)" $M EZKL_EVALH_R29=1 ;;
  merged)
    run "defaults" $M; run "MERGED_COMMITS=1" $M EZKL_PROVER_MERGED_COMMITS=1; run "SYNC_CALLS=1" $M EZKL_PROVER_SYNC_CALLS=1 ;;
  early)
    run "early (default)" $M; run "off" $M EZKL_PROVER_NO_EARLY_RANDOM=1
    run "early, LIB high AUX low" $M EZKL_HIP_PRIO_LIB=-1 EZKL_HIP_PRIO_AUX=1; run "off, LIB high AUX low" $M EZKL_PROVER_NO_EARLY_RANDOM=1 EZKL_HIP_PRIO_LIB=-1 EZKL_HIP_PRIO_AUX=1
    for C in "CIRCUIT=einsum K=20" "CIRCUIT=mlp K=17" "CIRCUIT=conv K=17"; do run "$C early" $C REPS=8; run "$C off" $C REPS=8 EZKL_PROVER_NO_EARLY_RANDOM=1; done ;;
  matrix)
    for E in early off; do
      if [ $E = off ]; then X="EZKL_PROVER_NO_EARLY_RANDOM=1"; else X="A=1"; fi
      for BIG in 1 2 4; do run "mlp20 $E BIG=$BIG" $X EZKL_MSM_GROUP_BIG=$BIG $M; run "einsum20 $E BIG=$BIG" $X EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=einsum K=20 REPS=8; done
    done
    for BIG in 1 4; do run "mlp22 early BIG=$BIG" EZKL_MSM_GROUP_BIG=$BIG CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=3; done ;;
  slots)
    for S in 2 3 4 6; do for BIG in 1 2 4; do run "SLOTS=$S BIG=$BIG" $M EZKL_MSM_SLOTS=$S EZKL_MSM_GROUP_BIG=$BIG; done; done
    run "mlp17 SLOTS=2" EZKL_MSM_SLOTS=2 CIRCUIT=mlp K=17 REPS=8; run "mlp17 SLOTS=4" EZKL_MSM_SLOTS=4 CIRCUIT=mlp K=17 REPS=8 ;;
  ntt29r)
    # pass radices under the radix-2^29 pass: two passes of 2^10 (4096-element tiles, one workgroup of 1024 threads per CU) against three of 2^7
    for X in 8 9 10; do echo "== EZKL_NTT_MAXR=$X"; (cd "$R" && EZKL_NTT_MAXR=$X timeout 300 python tools/ntt_ab.py 2>&1 | tail -4); done
    for X in 8 10; do run "mlp20 MAXR=$X" $M EZKL_NTT_MAXR=$X; done ;;
  sumscatter)
    # SHPLONK's two commitments reduce-scattered by point ranges against whole MSMs of the partials on every rank: 2 and 4 contexts on ONE
    # device (the contexts share its CUs: what shows here is the MSM work saved, not the exchange over xGMI)
    for W in 2 4; do for V in 0 1 0 1; do
      X="A=1"; [ $V = 1 ] && X="EZKL_PROVER_NO_SUM_SCATTER=1"
      (cd "$R" && env $X CONTEXTS=$(python -c "print(','.join(['0'] * $W))") CIRCUIT=mlp K=20 timeout 600 python tools/prove_group.py --pinned) 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); t = j['group_breakdown_seconds_max_over_contexts']
print('contexts $W NO_SUM_SCATTER=$V', 'one context', j['prove_seconds_one_context'], 'group', j['prove_seconds_group'], 'shplonk', t.get('shplonk'), 'evaluations', t.get('evaluations'),
      'same_as_one', j['same_bytes_as_one_context'], j['proof_sha256'], 'bytes received', [c.get('exchange_bytes_received') for c in j['per_context']])"
    done; done ;;
  taper)
    for T in taper equal; do
      if [ $T = equal ]; then X="EZKL_MSM_NO_TAPER=1"; else X="A=1"; fi
      run "mlp20 $T" $X $M; run "einsum20 $T" $X CIRCUIT=einsum K=20 REPS=8; run "mlp17 $T" $X CIRCUIT=mlp K=17 REPS=8; run "conv17 $T" $X CIRCUIT=conv K=17 REPS=8
      run "mlp22 $T" $X CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=25 REPS=3
    done ;;
  benchvar)
    for i in 1 2; do for S in 4 6; do
      (cd "$R" && EZKL_MSM_SLOTS=$S python bench.py --no-cpu-baseline) 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('run $i SLOTS=$S value %.4g ms/step %.4f acc %.4f msm_dev %.4f ntt_dev %.4f modmul29 %.4g copy %.0f' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['extra']['msm_device_ms'], j['extra']['ntt_device_ms'], j['extra']['modmul29_per_s'], j['extra']['hbm_copy_GBs']))"
    done; done
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6 ;;
  skew)
    # witness-shaped columns (round 5): the lane-length floor and the cut count above which a bucket takes the whole-workgroup heavy path
    run "defaults (LMIN=8 SPAN=16)" $M
    for L in 16 32; do run "LMIN=$L" $M EZKL_MSM_LMIN=$L; done
    for S in 32 64; do run "SPAN=$S" $M EZKL_MSM_SPAN=$S; done
    run "LMIN=16 SPAN=32" $M EZKL_MSM_LMIN=16 EZKL_MSM_SPAN=32 ;;
  skew2)
    run "defaults (LMIN=8 SPAN=16)" $M
    for S in 4 8; do run "SPAN=$S" $M EZKL_MSM_SPAN=$S; done
    run "LMIN=16 SPAN=8" $M EZKL_MSM_LMIN=16 EZKL_MSM_SPAN=8
    run "LMIN=24" $M EZKL_MSM_LMIN=24
    run "defaults again" $M ;;
  msmdebug)
    (cd "$R" && EZKL_MSM_DEBUG=1 CIRCUIT=mlp K=20 REPS=1 timeout 300 python tools/prove_bench.py --pinned) 2>&1 | grep "msm batch" | tail -12 ;;
  *) echo "unknown experiment '$1'"; exit 2 ;;
esac
