#!/bin/bash
# ThreadSanitizer over the kernel sources (CPU only; GPU sanitizers are not available): the emulator is built with -fsanitize=thread and
# run with EMU_WAVES, i.e. the four waves of every workgroup as host threads with an atomic barrier per phase, so TSAN reports every pair of
# conflicting LDS / global accesses that no barrier orders.  Usage: bash tests/emu/tsan.sh [seed]    (about a minute)
# Known, intended report: kernels/inter.h me_search_program `consider` reads s.best[] plainly as a filter before the atomic minimum.
set -e
here=$(cd "$(dirname "$0")" && pwd)
so=/tmp/libkernel_emu_tsan.so
g++ -std=c++17 -O1 -g -fPIC -shared -w -pthread -fsanitize=thread -o $so $here/emu.cpp
EMU_WAVES=${1:-3} TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" LD_PRELOAD=$(gcc -print-file-name=libtsan.so) python3 $here/tsan_case.py $so > /tmp/tsan.log 2>&1 || true
grep "same" /tmp/tsan.log
echo "reports: $(grep -c 'WARNING: ThreadSanitizer' /tmp/tsan.log)  (full log: /tmp/tsan.log)"
grep "SUMMARY" /tmp/tsan.log | cut -c1-200 | sort | uniq -c | sort -rn
