#!/usr/bin/env python3
"""VERDICT r03 item 3 -- "a faster field product": the measured table behind NOTEBOOK.md §4.0.

  * issue probes (8 independent chains per lane, 4096 iterations): v_mad_u64_u32, v_fma_f64, the two interleaved 4 + 4 (co-issue would
    show as more lane-ops/s than either alone), v_lshl_add_u64 (one-instruction 64-bit add), v_add_f64;
  * complete Montgomery products per second: radix-2^29 v_mad_u64_u32 (what every kernel ships), radix-2^32 (round 1), and the 5 x 52-bit
    FP64-pipe product (Emmart-style fma_rz hi / lo splitting, 345 instructions per product: 110 v_fma_f64 + 71 v_add_f64 + 99
    v_lshl_add_u64 + bit moves) -- each at 1..8 waves per SIMD and at the occupancy its registers allow;
  * exactness: the FP64 product against a 64-bit integer restatement on 2 x (CUs x 16 x 256) random operand pairs per run, `CHECK_RUNS`
    runs with fresh operands are not needed -- the kernel derives its operands from distinct words per thread."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
print("exactness: dfma52 mismatches vs the integer restatement:", B.ubench("dfma52_check"), " radix-2^29 mismatches:", B.ubench("modmul29_check"))
print("%-26s %s" % ("probe (lane-ops/s)", "  ".join("o%-9s" % o for o in ("1", "2", "4", "8", "max"))))
for name in ("mix_mad", "mix_dfma", "mix_both", "mix_add64", "mix_dadd"):
    print("%-26s %s" % (name, "  ".join("%.3e" % B.ubench(name + o) for o in ("_o1", "_o2", "_o4", "_o8", ""))))
print("%-26s %s" % ("product (products/s)", "  ".join("o%-9s" % o for o in ("1", "2", "3", "4", "8", "max"))))
for name in ("modmul29", "dfma52", "modmul"):
    print("%-26s %s" % (name, "  ".join("%.3e" % B.ubench(name + o) for o in ("_o1", "_o2", "_o3", "_o4", "_o8", ""))))
