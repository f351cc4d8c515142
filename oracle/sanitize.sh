#!/bin/bash
# The oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU build
# only): builds oracle/_build/libnerfacc_oracle_san.so and runs the CPU tests that call into it.  The sanitizer runtime
# has to be the first library of the process, hence the preload; leak checking is off (the interpreter's own).
#   oracle/sanitize.sh            -> exit code of pytest
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
export ORACLE_SANITIZE=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
LD_PRELOAD="$ASAN $UBSAN" python -m pytest tests/test_oracle_golden.py tests/test_host_logic.py -x -q -m "not gpu" "$@"
