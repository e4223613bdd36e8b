#!/bin/bash
# The host-only templates (BKLDLT.h, the line-search state machines, Param.h, BFGSMat.h's solve_PtBP against a mock of the C ABI)
# under -fsanitize=address,undefined: tests/test_host_logic_cpu.py with its helper libraries rebuilt with the sanitizers and
# libasan preloaded into the interpreter.  CPU only (GPU AddressSanitizer is not available on this pool).  Once per round;
# the log line is committed as profiles/rN_host_sanitizers.txt.
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
LD_PRELOAD="$ASAN:$UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  LBFGSX_TEST_SANITIZE=1 python -m pytest tests/test_host_logic_cpu.py -q -p no:cacheprovider 2>&1 | tail -15
