#!/bin/bash
# tools/kernel_pmc.sh <tag> -- the SQ instruction / stall counters of tools/ntt_pmc.sh for ANY kernel of a probe, plus its durations:
#   KERNELS="msm_partition_kernel msm_hist_kernel" PROBE=pmc_probe.py bash tools/run.sh <tag> sh:kernel_pmc.sh   -> gpurun_out/<tag>_kernel_pmc.txt
# Counters in their own rocprofv3 passes (8 SQ slots each), the kernel trace in a third run.
TAG=${TAG:-${1:-kpmc}}
KERNELS=${KERNELS:-"msm_partition_kernel msm_hist_kernel msm_binsort_kernel msm_accumulate_kernel"}; PROBE=${PROBE:-pmc_probe.py}
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out/${TAG}_kpmc"; OUT="$R/gpurun_out/${TAG}_kernel_pmc.txt"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/a" -- python "$R/tools/$PROBE" > "$O/a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/b" -- python "$R/tools/$PROBE" > "$O/b.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/c" -- python "$R/tools/$PROBE" > "$O/c.log" 2>&1
A=$(find "$O/a" -name '*counter_collection.csv' | head -1); B=$(find "$O/b" -name '*counter_collection.csv' | head -1); C=$(find "$O/c" -name '*kernel_stats.csv' | head -1)
: > "$OUT"
for K in $KERNELS; do KERNEL=$K python "$R/tools/ntt_pmc_reduce.py" "$A" "$B" >> "$OUT" 2>&1; echo >> "$OUT"; done
echo "== kernel durations (own run, no counters)" >> "$OUT"; head -30 "$C" | cut -d, -f1-8 >> "$OUT"
tail -3 "$O/a.log" "$O/b.log" "$O/c.log" >> "$OUT"
rm -rf "$O"; cat "$OUT"
