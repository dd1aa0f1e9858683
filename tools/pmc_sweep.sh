#!/bin/bash
# PMC counters of the quotient-sweep kernel of the MLP proof (rocprofv3 --pmc, kernel rows of evalh_jit with the most instructions)
#   usage (GPU box): bash tools/pmc_sweep.sh <tag> [env assignments...]
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
O="$R/gpurun_out/${TAG}"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp EZKL_HIP_CACHE_DIR=off
env "$@" CIRCUIT=mlp K=${K:-18} REPS=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d "$O/pmc" -- python "$R/tools/prove_bench.py" --native --pinned > "$O/log.txt" 2>&1
F=$(find "$O/pmc" -name '*counter_collection.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "evalh_jit" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
best = max(agg.values(), key=lambda d: d.get("SQ_INSTS_VALU", 0))
print({k: round(v) for k, v in best.items()})
PY
rm -rf "$O/pmc"
