#!/bin/bash
# duration of the quotient-sweep KERNEL (rocprofv3 kernel trace; the longest evalh_jit dispatch) for one code-generation configuration
#   usage (GPU box): bash tools/sweep_kernel_time.sh ENV=VAL,ENV=VAL
R=$(cd "$(dirname "$0")/.." && pwd)
O="$R/gpurun_out/skt"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp EZKL_HIP_CACHE_DIR=off
env $(echo ${1:-X=1} | tr ',' ' ') CIRCUIT=mlp K=${K:-18} REPS=2 rocprofv3 --kernel-trace -d "$O" -- python "$R/tools/prove_bench.py" --native --pinned > /dev/null 2>&1
DB=$(find "$O" -name '*.db' | head -1)
python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
d = sorted((e - s) / 1e3 for n, s, e in cur.execute("select %s, start, end from kernels" % name_col) if "evalh_jit" in n)
print("evalh_jit dispatches %d, longest three (us): %s" % (len(d), [round(x) for x in d[-3:]]))
PY
rm -rf "$O"
