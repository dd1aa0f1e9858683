"""The radix-2^29 NTT pass (ezkl_amd/csrc/ntt.hip, ntt_pass29_kernel) as an integer model, tools/ntt29_model.py: the lazily reduced DIT
butterfly with every 32- / 64-bit register checked, against the plain DFT over Fr.  Host logic only: the kernel itself is compared with
the oracle by tests/test_gpu_ntt.py on the GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import ntt29_model as M


def test_column_transforms_stay_in_range_and_match_the_dft():
    worst = M.selftest(seed=3, log_rs=(1, 2, 3, 4, 6, 7, 8))
    # the bound the kernel's comment states: 4p per stage on top of a 256-bit input plus the first stage's 8p
    assert worst["max_value_over_p"] < 54
    # three stages between carry passes would also stay inside 32-bit limbs (the kernel uses two: register pressure)
    assert M.selftest(seed=4, log_rs=(3, 6, 7), group=3)["max_value_over_p"] < 54


def test_constants_are_the_generated_ones():
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ezkl_amd", "csrc", "montmul29_gen.hpp")).read()
    fr = src[src.index("struct Fr29C"):]
    def row(tag):
        line = [l for l in fr.splitlines() if tag in l][0]
        return [int(x.rstrip("u"), 16) for x in line[line.index("{") + 1:line.index("}")].split(", ")]
    assert row("// 2^261 - 1 p") == M.CSUB_P
    assert row("// 4 p") == M.SUBC[4] and row("// 8 p") == M.SUBC[8]
    assert row("P[9]") == M.limbs29(M.P)
