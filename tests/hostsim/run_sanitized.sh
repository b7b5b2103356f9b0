#!/bin/bash
# The CPU test suite (-m "not gpu") with the host engine built under a sanitizer:
#     tests/hostsim/run_sanitized.sh asan [pytest args]      AddressSanitizer + UBSan over engine.cpp / cc.cpp / capi.cpp / fcidump.cpp
#     tests/hostsim/run_sanitized.sh tsan [pytest args]      ThreadSanitizer (the threaded FCIDUMP parser: tests/test_fcidump_hf.py)
# The interpreter is not instrumented, so the runtime is preloaded — with libstdc++ next to it, or the interception of
# __cxa_throw aborts at the first C++ exception that crosses the C-ABI.  GPU sanitizers are not available on this pool:
# this is where the host side of the product gets its sanitizer coverage.
set -e
kind=${1:?asan or tsan}; shift
here=$(cd "$(dirname "$0")" && pwd)
make -s -C "$here" "$kind"
case $kind in
  asan) rt=$(g++ -print-file-name=libasan.so); export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ;;
  tsan) rt=$(g++ -print-file-name=libtsan.so); export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66 suppressions=$here/tsan.supp" ;;
  *) echo "asan or tsan"; exit 2 ;;
esac
export PYMES_HOSTSIM_LIBRARY="$here/_build/$kind/libpymes_hostsim.so"
cd "$here/../.."
if [ $# -eq 0 ]; then set -- tests -m "not gpu" -x -q; fi
LD_PRELOAD="$rt $(g++ -print-file-name=libstdc++.so.6)" python -m pytest "$@"
