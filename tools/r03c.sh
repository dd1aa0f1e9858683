#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd); O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for Q in 8 16 32; do
  GPU_MAX_HW_QUEUES=$Q CIRCUIT=mlp K=20 REPS=5 timeout 600 python "$R/tools/prove_bench.py" --pinned > "$O/r03c_q$Q.log" 2>&1
  echo "HWQ=$Q $(tail -1 $O/r03c_q$Q.log | grep -o '"prove_seconds_gpu_runs": [^]]*]') $(tail -1 $O/r03c_q$Q.log | grep -o '"prove_breakdown_seconds": {[^}]*}')"
done
GPU_MAX_HW_QUEUES=16 CIRCUIT=mlp K=20 REPS=3 timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -d "$O/r03c_prove" -- python "$R/tools/prove_bench.py" --pinned > "$O/r03c_prove.log" 2>&1
DB=$(find "$O/r03c_prove" -name '*.db' | head -1)
python "$R/tools/gantt.py" "$DB" 100 250 > "$O/r03c_prove_gantt.txt" 2>&1
python - "$DB" > "$O/r03c_schema.txt" 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select * from regions order by start desc limit 3"))
print([r[1] for r in cur.execute("pragma table_info(regions)")])
for r in rows: print(r)
for v in ("region_args", "events_args"):
    print(v, [r[1] for r in cur.execute("pragma table_info(%s)" % v)])
    for r in cur.execute("select * from %s limit 6" % v): print("  ", r)
print("copies:")
t1 = max(r[0] for r in cur.execute("select max(end) from kernels"))
for s, e, sz, nm in cur.execute("select start, end, size, name from memory_copies where end > ? and size > 1000000 order by start", (t1 - 100e6,)):
    print("  %.3f ms  %.1f us  %d B  %s" % ((s - (t1 - 100e6)) / 1e6, (e - s) / 1e3, sz, nm))
PY
rm -rf "$O/r03c_prove"
cat "$O/r03c_schema.txt" | head -60
