#!/usr/bin/env python3
"""per launch of ntt_pass_kernel (grouped by grid size = by transform size / pass shape): the SQ counters of tools/ntt_pmc.sh, and the ratios the
VERDICT asks for -- Montgomery products per VALU instruction, LDS instructions per VALU instruction, share of issue cycles stalled on LDS."""
import csv, os, sys
from collections import defaultdict, OrderedDict
KERNEL = os.environ.get("KERNEL", "ntt_pass")          # substring of the kernel name (tools/kernel_pmc.sh sets it)
rows = defaultdict(lambda: defaultdict(list))          # (grid, wg) -> counter -> values
for path in sys.argv[1:3]:
    for r in csv.DictReader(open(path)):
        if KERNEL not in r["Kernel_Name"]:
            continue
        key = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r.get("LDS_Block_Size", 0) or 0))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU",
         "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
print(KERNEL + ", SQ counters per launch (mean over the launches of one shape; SQ_*_CYCLES / ACTIVE / WAIT in quad-cycles summed over waves)")
for key in sorted(rows):
    c = {n: (sum(rows[key][n]) / len(rows[key][n]) if rows[key][n] else float("nan")) for n in names}
    launches = max(len(v) for v in rows[key].values())
    print("\ngrid %d x wg %d, lds %d B, %d launches" % (key[0], key[1], key[2], launches))
    for n in names:
        print("  %-22s %.4g" % (n, c[n]))
    v = c["SQ_INSTS_VALU"]
    if v == v and v:
        print("  -> LDS instr / VALU instr      %.4f" % (c["SQ_INSTS_LDS"] / v))
        print("  -> VMEM instr / VALU instr     %.4f" % ((c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / v))
        print("  -> VALU instr per wave         %.1f" % (v / c["SQ_WAVES"]))
    a = c["SQ_ACTIVE_INST_ANY"]
    if a == a and a:
        tot = c["SQ_ACTIVE_INST_ANY"] + c["SQ_WAIT_INST_ANY"] + c["SQ_WAIT_ANY"]
        print("  -> of wave cycles: issuing %.1f %% (VALU %.1f %%, LDS %.1f %%), issue-stalled %.1f %% (on LDS %.1f %%), parked (waitcnt / barrier) %.1f %%"
              % (100 * a / tot, 100 * c["SQ_ACTIVE_INST_VALU"] / tot, 100 * c["SQ_ACTIVE_INST_LDS"] / tot, 100 * c["SQ_WAIT_INST_ANY"] / tot,
                 100 * c["SQ_WAIT_INST_LDS"] / tot, 100 * c["SQ_WAIT_ANY"] / tot))
        print("  -> LDS bank-conflict cycles / LDS active cycles %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c["SQ_LDS_IDX_ACTIVE"] else float("nan")))
