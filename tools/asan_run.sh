#!/bin/bash
# The host prover (csrc/prover/prover.cpp: key files, circuit blobs, transcripts, pairing) under AddressSanitizer + UBSan.
#   tools/asan_run.sh [pytest args]      default: the CPU tests that load libezkl_prover.so (malformed keys / blobs included)
# On a GPU box the same command with `-m gpu` runs the provers' GPU tests under the sanitizers (the HIP library itself is not instrumented).
# The log goes to profiles/ (tracked).  detect_leaks=0: CPython itself "leaks" at exit; the library's own frees are covered by the GPU suite.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C "$R/ezkl_amd/csrc" asan > /dev/null
ASAN_LIB=$(gcc -print-file-name=libasan.so)
UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
ARGS=("$@")
if [ ${#ARGS[@]} -eq 0 ]; then
  ARGS=(tests/test_native_prover.py tests/test_execute.py tests/test_gen_witness.py tests/test_bench_cache.py tests/test_contexts_cpu.py tests/test_capi.py -m "not gpu")
fi
cd "$R"
LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  EZKL_PROVER_LIB="$R/ezkl_amd/libezkl_prover_asan.so" python -m pytest -x -q "${ARGS[@]}"
