#!/usr/bin/env python3
"""Latency probes for chains of dependent point additions (ezkl_hip_ubench "ecadd<iters><mode>"; NOTEBOOK.md §4.1):
w = one wave, q = one wave per CU, h = one wave per SIMD, f = four waves per SIMD; t/u/v/x = tree variants that separate the
cost of lane exchange from the cost of a partial EXEC mask (the finding: masked-off lanes make the chain 2.3x slower)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
for w in ["ecadd0w","ecadd1w","ecadd17w","ecadd65w","ecadd1q","ecadd17q","ecadd1h","ecadd17h","ecadd65h","ecaddtw","ecaddth","ecadduw","ecadduh","ecaddvw","ecaddvh","ecaddxw","ecaddxh","ecadd6w","ecadd6h","ecadd0f","ecadd1f","ecadd17f"]:
    print(w, [round(B.ubench(w),1) for _ in range(2)])
