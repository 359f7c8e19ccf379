#!/bin/bash
# Sanitizer runs of everything that has a CPU build (GPU sanitizers do not exist on this pool): bash tests/sanitize_cpu.sh
#  1. kernel sources stepped on the CPU (tests/emu) under ASAN + UBSAN, sequential lanes
#  2. the same under TSAN with four wave threads and real barriers (EMU_WAVES)
#  3. oracle + decoder (oracle/) and the host coder (bitstream.cpp, api_host.cpp built with g++) under ASAN + UBSAN: their pytest files
# Every library is built into /tmp; nothing in the tree changes.  Exit code 0 = no report.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
gccdir=$(dirname "$(gcc -print-file-name=libasan.so)")
asan="$gccdir/libasan.so:$gccdir/libubsan.so"
tmp=$(mktemp -d /tmp/mihevc_san.XXXXXX)
fail=0
echo "== 1. stepped kernels, ASAN + UBSAN"
g++ -std=c++17 -O1 -g -fPIC -shared -w -pthread -fsanitize=address,undefined -o $tmp/libemu_asan.so tests/emu/emu.cpp
EMU_LIB=$tmp/libemu_asan.so LD_PRELOAD=$asan ASAN_OPTIONS=detect_leaks=0 python3 tests/sanitize_cases.py > $tmp/1.log 2>&1 || fail=1
grep -E "^ok|runtime error|ERROR: AddressSanitizer" $tmp/1.log | sort | uniq -c
grep -qE "runtime error|ERROR: AddressSanitizer" $tmp/1.log && fail=1
echo "== 2. stepped kernels on four wave threads, TSAN"
g++ -std=c++17 -O1 -g -fPIC -shared -w -pthread -fsanitize=thread -o $tmp/libemu_tsan.so tests/emu/emu.cpp
EMU_WAVES=7 EMU_LIB=$tmp/libemu_tsan.so LD_PRELOAD=$gccdir/libtsan.so TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" python3 tests/sanitize_cases.py > $tmp/2.log 2>&1 || fail=1
grep -E "^ok|SUMMARY: ThreadSanitizer" $tmp/2.log | cut -c1-160 | sort | uniq -c
grep -q "WARNING: ThreadSanitizer" $tmp/2.log && fail=1
echo "== 3. oracle + decoder + host coder, ASAN + UBSAN"
gcc -O1 -g -fPIC -shared -fsanitize=address,undefined -o $tmp/liboracle.so oracle/hevc_oracle.c oracle/hevc_dec.c oracle/hevc_dec_recon.c -lm
python3 - "$tmp" <<'PY'
import subprocess, sys
sys.path.insert(0, '.')
from hevc_amd import _lib
tmp = sys.argv[1]
have = {l.split()[-1] for l in subprocess.run(['nm', '-D', '--defined-only', 'hevc_amd/libmihevc.so'], capture_output=True, text=True).stdout.splitlines()}
host = subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-fPIC', '-shared', '-fsanitize=address,undefined', '-o', tmp + '/host_only.so', 'hevc_amd/csrc/bitstream.cpp',
                       'hevc_amd/csrc/api_host.cpp', '-Ihevc_amd/csrc', '-Iinclude', '-lpthread'], capture_output=True, text=True)
assert host.returncode == 0, host.stderr[-2000:]
got = {l.split()[-1] for l in subprocess.run(['nm', '-D', '--defined-only', tmp + '/host_only.so'], capture_output=True, text=True).stdout.splitlines()}
open(tmp + '/stubs.cpp', 'w').write('extern "C" {\n' + ''.join(f'int {n}() {{ return -19; }}\n' for n in _lib.EXPORTS if n not in got) + '}\n')   # device entry points: ENODEV
r = subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-fPIC', '-shared', '-fsanitize=address,undefined', '-o', tmp + '/libmihevc_host_asan.so', 'hevc_amd/csrc/bitstream.cpp',
                    'hevc_amd/csrc/api_host.cpp', tmp + '/stubs.cpp', '-Ihevc_amd/csrc', '-Iinclude', '-lpthread'], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-2000:]
PY
cat > $tmp/run3.py <<PY
import sys
from pathlib import Path
sys.path.insert(0, '$root')
import oracle.oracle as O
O.build = lambda force=False: Path('$tmp/liboracle.so')
import pytest
sys.exit(int(pytest.main(['$root/tests/test_oracle_kat.py', '$root/tests/test_bitstream_cpu.py', '$root/tests/test_sliced_cpu.py', '$root/tests/test_decoder_second_opinion.py', '-x', '-q', '-p', 'no:cacheprovider'])))
PY
MIHEVC_LIBRARY=$tmp/libmihevc_host_asan.so LD_PRELOAD=$asan ASAN_OPTIONS=detect_leaks=0 python3 $tmp/run3.py > $tmp/3.log 2>&1 || fail=1
tail -2 $tmp/3.log
grep -E "runtime error|ERROR: AddressSanitizer" $tmp/3.log | sort | uniq -c
grep -qE "runtime error|ERROR: AddressSanitizer" $tmp/3.log && fail=1
rm -rf $tmp
[ $fail = 0 ] && echo "sanitizers: clean" || echo "sanitizers: REPORTS ABOVE"
exit $fail
