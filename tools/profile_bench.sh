#!/bin/bash
# Runs on the GPU box (through gpurun): bench line + rocprofv3 kernel trace + the two HBM-traffic PMC passes of the SAME command.
# (--profile-stages 1 in the counter passes: bench.py then runs no extra per-stage step, so a pass is exactly ONE encode of the clip)
# Usage: bash tools/profile_bench.sh <tag>      -> gpurun_out/<tag>/{bench.json,kernel_stats.txt,kt_kernel_stats.csv,traffic.json}
set -e
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 5 > $out/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o f --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --profile-stages 1 > /dev/null
rocprofv3 --pmc WRITE_SIZE -d $out/write -o w --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --profile-stages 1 > /dev/null
cd $root
python3 tools/prof_summary.py stats $out/kt > $out/kernel_stats.txt
cp $(find $out/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
python3 tools/prof_summary.py traffic $out/fetch $out/write $out/traffic.json > /dev/null
rm -rf $out/kt $out/fetch $out/write
cat $out/bench.json; cat $out/kernel_stats.txt; cat $out/traffic.json
