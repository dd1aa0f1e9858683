#!/bin/bash
# tools/run.sh <tag> <job> [<job> ...] -- ONE parametrised runner for the GPU box (replaces round 3's 35 per-run tools/r03*.sh wrappers).
# Every job writes gpurun_out/<tag>_<job>*; copy what should be judged into profiles/ (profiles/README.md lists what each file is).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/run.sh r04a suite bench'
# jobs:
#   suite        pytest -m gpu with --durations=40 and the memory report           -> <tag>_pytest_gpu.log
#   bench        python bench.py (the driver's default command), wall time logged   -> <tag>_bench.log
#   bench2       the bench contract at world 2 on ONE device (gloo plumbing test)   -> <tag>_bench2.log
#   profile      rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline`   -> <tag>_kernel_stats.txt, <tag>_bench_profiled.log
#   pmc          FETCH_SIZE / WRITE_SIZE passes over the 2^20 MSM + 2^22 NTT        -> <tag>_pmc_traffic.json
#   pmcprove     the same two passes over one k = 20 MLP proof                      -> <tag>_pmc_prove.json
#   nttpmc       SQ_* instruction counters of ntt_pass_kernel at 2^20 and 2^22      -> <tag>_ntt_pmc.txt
#   mlp20        CIRCUIT=mlp K=20 REPS=8 tools/prove_bench.py --pinned              -> <tag>_mlp_k20.log
#   mlp20cold    the same with --cold                                               -> <tag>_mlp_k20_cold.log
#   timeline     kernel + copy + marker trace of one k = 20 MLP proof               -> <tag>_timeline.txt, <tag>_hosttrace.txt, <tag>_gantt.txt
#   msmcols      the twelve advice columns committed one at a time                  -> <tag>_msm_columns_serial.txt
#   ubench       Montgomery-product micro-benchmarks (mad64 / radix-2^29 / DFMA)    -> <tag>_ubench.log
#   keygen       stage table of ezkl_prover_keygen at k = 20                        -> <tag>_keygen.log
#   group        prover group, 2 and 4 contexts on one device, k = 20 MLP            -> <tag>_group{2,4}.json
#   multi2       two owner-mode ranks (gloo) sharing the device, k = 20 MLP          -> <tag>_multi2.json
#   k22          K=22 MLP_BLOCKS=5 MLP_FILL=25 proof + HBM high-water (opt-in size)  -> <tag>_mlp_k22.log
#   k22cold      the same with the cold one-shot (writes 68 GB of artefacts to /tmp) -> <tag>_mlp_k22_cold.log
#   ab:<name>    an A/B experiment of tools/ab.sh (group, circuits, hwq, prio, evalh, ...)  -> <tag>_ab_<name>.log
#   sh:<file>    bash tools/<file>                                                    -> <tag>_<file>.log
#   tests:<a,b>  pytest -m gpu on the listed files                                  -> <tag>_pytest_subset.log
#   py:<file>    python <file> (a probe under tools/)                                -> <tag>_<file>.log
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
TAG=${1:?tag}; shift
export TMPDIR=/tmp
for JOB in "$@"; do
  T0=$(date +%s)
  case "$JOB" in
    suite)
      (cd "$R" && timeout ${SUITE_TIMEOUT:-900} python -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider) > "$O/${TAG}_pytest_gpu.log" 2>&1
      echo "pytest rc=$?" >> "$O/${TAG}_pytest_gpu.log"; tail -60 "$O/${TAG}_pytest_gpu.log" ;;
    bench)
      TB=$(date +%s)
      (cd "$R" && timeout ${BENCH_TIMEOUT:-600} python bench.py ${BENCH_ARGS:-}) > "$O/${TAG}_bench.log" 2> "$O/${TAG}_bench.err"
      echo "bench.py rc=$? wall $(( $(date +%s) - TB )) s" >> "$O/${TAG}_bench.log"
      grep '^{"metric' "$O/${TAG}_bench.log" | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('line %d bytes  value %.4g %s  ms/step %.4f  roofline.frac %.4f  ntt %.4g el/s (inverse %.4g, coset %.4g)  W %s' % (len(json.dumps(j, separators=(',', ':'))), j['value'], j['unit'], j['ms_per_step'], r['frac'], r['ntt']['elems_per_s'], r['ntt'].get('inverse_elems_per_s') or 0, r['ntt'].get('coset_2p20_to_2p22_elems_per_s') or 0, r.get('msm_witness_like')))
print('prove_seconds_k20_mlp', j.get('prove_seconds_k20_mlp')); print('others', j.get('prove_other_circuits'), 'skipped', j.get('prove_skipped'), 'errors', j.get('errors'))"
      cp "$R/bench_full.json" "$O/${TAG}_bench_full.json" 2>/dev/null
      tail -1 "$O/${TAG}_bench.log" ;;
    bench2)
      (cd "$R" && timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --share-device) > "$O/${TAG}_bench2.log" 2>&1
      grep '^{"metric' "$O/${TAG}_bench2.log" | cut -c1-600 ;;
    profile)
      bash "$R/tools/profile_bench.sh" "$TAG" ;;
    pmc)
      bash "$R/tools/pmc_run.sh" "$TAG" ;;
    pmcprove)
      bash "$R/tools/pmc_prove.sh" "$TAG" ;;
    nttpmc)
      bash "$R/tools/ntt_pmc.sh" "$TAG" ;;
    mlp20)
      (cd "$R" && CIRCUIT=mlp K=20 REPS=${REPS:-8} timeout 600 python tools/prove_bench.py --pinned) > "$O/${TAG}_mlp_k20.log" 2>&1
      tail -1 "$O/${TAG}_mlp_k20.log" | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['prove_seconds_gpu_runs'], j['prove_breakdown_seconds'], 'keygen', j['keygen_seconds_gpu'], 'hbm', j.get('hbm_in_use_gib_after_prove'), j.get('hbm_pool_high_water_gib'), j['proof_sha256'])" ;;
    mlp20cold)
      (cd "$R" && CIRCUIT=mlp K=20 REPS=3 timeout 900 python tools/prove_bench.py --pinned --cold) > "$O/${TAG}_mlp_k20_cold.log" 2>&1
      tail -1 "$O/${TAG}_mlp_k20_cold.log" | cut -c1-1500 ;;
    timeline)
      (cd /tmp && CIRCUIT=mlp K=20 REPS=3 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d "$O/${TAG}_prove" -- python "$R/tools/prove_bench.py" --pinned) > "$O/${TAG}_prove.log" 2>&1
      DB=$(find "$O/${TAG}_prove" -name '*.db' | head -1)
      python "$R/tools/hosttrace.py" "$DB" ${WINDOW_MS:-93} 120 > "$O/${TAG}_hosttrace.txt" 2>&1
      python "$R/tools/timeline.py" "$DB" ${WINDOW_MS:-93} > "$O/${TAG}_timeline.txt" 2>&1
      python "$R/tools/gantt.py" "$DB" ${WINDOW_MS:-93} 250 > "$O/${TAG}_gantt.txt" 2>&1
      rm -rf "$O/${TAG}_prove"; head -30 "$O/${TAG}_timeline.txt" ;;
    msmcols)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$O/${TAG}_msmcols" -- python "$R/tools/msm_columns_profile.py" run) > "$O/${TAG}_msmcols.log" 2>&1
      DB=$(find "$O/${TAG}_msmcols" -name '*.db' | head -1)
      python "$R/tools/msm_columns_profile.py" reduce "$DB" "$O/${TAG}_msmcols.log" > "$O/${TAG}_msm_columns_serial.txt" 2>&1
      rm -rf "$O/${TAG}_msmcols"; cat "$O/${TAG}_msm_columns_serial.txt" ;;
    ubench)
      (cd "$R" && timeout 300 python tools/ubench_products.py) > "$O/${TAG}_ubench.log" 2>&1; cat "$O/${TAG}_ubench.log" ;;
    keygen)
      (cd "$R" && EZKL_PROVER_KEYGEN_TIMING=1 CIRCUIT=mlp K=${KEYGEN_K:-20} REPS=1 timeout 900 python tools/prove_bench.py --pinned) > "$O/${TAG}_keygen.log" 2>&1
      grep -i "keygen" "$O/${TAG}_keygen.log" | head -40 | cut -c1-400 ;;
    group)
      for W in 2 4; do (cd "$R" && CONTEXTS=$(python -c "print(\",\".join([\"0\"]*$W))") CIRCUIT=mlp K=20 timeout 600 python tools/prove_group.py --pinned) > "$O/${TAG}_group$W.json" 2> "$O/${TAG}_group$W.err"; tail -1 "$O/${TAG}_group$W.json" | cut -c1-700; done ;;
    multi2)
      (cd "$R" && CIRCUIT=mlp K=20 REPS=3 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 tools/prove_multi.py --pinned --gloo --share-device) > "$O/${TAG}_multi2.json" 2> "$O/${TAG}_multi2.err"
      tail -1 "$O/${TAG}_multi2.json" | cut -c1-900 ;;
    k22)
      (cd "$R" && CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=${K22_FILL:-25} REPS=2 timeout 1500 python tools/prove_bench.py --pinned) > "$O/${TAG}_mlp_k22.log" 2>&1
      tail -1 "$O/${TAG}_mlp_k22.log" | cut -c1-1500 ;;
    ab:*)
      F=${JOB#ab:}; bash "$R/tools/ab.sh" "$F" > "$O/${TAG}_ab_${F}.log" 2>&1; tail -40 "$O/${TAG}_ab_${F}.log" ;;
    sh:*)
      F=${JOB#sh:}; bash "$R/tools/$F" > "$O/${TAG}_${F%.sh}.log" 2>&1; tail -40 "$O/${TAG}_${F%.sh}.log" ;;
    tests:*)
      F=${JOB#tests:}; (cd "$R" && timeout ${SUITE_TIMEOUT:-900} python -m pytest $(echo "$F" | tr ',' ' ') -m gpu -q --durations=15 -p no:cacheprovider) > "$O/${TAG}_pytest_subset.log" 2>&1
      echo "pytest rc=$?" >> "$O/${TAG}_pytest_subset.log"; tail -40 "$O/${TAG}_pytest_subset.log" ;;
    k22cold)
      (cd "$R" && CIRCUIT=mlp K=22 MLP_BLOCKS=5 MLP_FILL=${K22_FILL:-25} REPS=2 EZKL_COLD_GAP_S=${EZKL_COLD_GAP_S:-10} EZKL_COLD_DIR=${EZKL_COLD_DIR:-/tmp} timeout 1800 python tools/prove_bench.py --pinned --cold) > "$O/${TAG}_mlp_k22_cold.log" 2>&1
      tail -1 "$O/${TAG}_mlp_k22_cold.log" | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['prove_seconds_gpu_runs'], 'cold', json.dumps(j.get('cold'))[:1500])"; df -h /tmp | tail -1 ;;
    py:*)
      F=${JOB#py:}; (cd "$R" && timeout ${PY_TIMEOUT:-600} python "tools/$F") > "$O/${TAG}_${F%.py}.log" 2>&1; tail -40 "$O/${TAG}_${F%.py}.log" ;;
    *) echo "unknown job $JOB" ;;
  esac
  echo "== $JOB: $(( $(date +%s) - T0 )) s"
done
