#!/bin/bash
# SQ instruction / stall counters of ntt_pass_kernel at 2^20 and 2^22 ("count, then cut") -> gpurun_out/<tag>_ntt_pmc.txt
# Counters in their own rocprofv3 passes (no kernel trace beside them), 8 SQ slots per pass.
TAG=${1:-nttpmc}
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out/${TAG}_nttpmc"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$O/a" -- python "$R/tools/ntt_pmc_probe.py" > "$O/a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/b" -- python "$R/tools/ntt_pmc_probe.py" > "$O/b.log" 2>&1
A=$(find "$O/a" -name '*counter_collection.csv' | head -1); B=$(find "$O/b" -name '*counter_collection.csv' | head -1)
python "$R/tools/ntt_pmc_reduce.py" "$A" "$B" > "$R/gpurun_out/${TAG}_ntt_pmc.txt" 2>&1
tail -5 "$O/a.log" "$O/b.log" >> "$R/gpurun_out/${TAG}_ntt_pmc.txt"
rm -rf "$O"; cat "$R/gpurun_out/${TAG}_ntt_pmc.txt"
