#!/bin/bash
# PMC passes for the per-CTU kernels on the BENCH command itself (bench.py's 300-frame 1080p clip, one timed step): bash tools/pmc_kernels.sh <tag>
# Each --pmc group is its own rocprofv3 run (never combined with traces); summary -> gpurun_out/<tag>/pmc_summary.txt
set -e
tag=${1:-pmc}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $out/p$i -o p --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --profile-stages 1 > $out/run$i.log 2>&1 || { tail -5 $out/run$i.log; exit 1; }
done
cd $root
{ echo "# mean per dispatch of the SQ counters rocprofv3 wrote for each kernel (three separate --pmc passes of: python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras)."
  echo "# Every value is the MEAN PER DISPATCH (not the sum over the run): the SQ block of a subset of the shader engines is sampled, so"
  echo "# absolute values are a sample.  Read RATIOS of counters of one pass, e.g. SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both in units of 4 cycles) = share of a"
  echo "# resident wave's life spent issuing VALU work; x waves per SIMD (occupancy) = share of SIMD issue slots: pmc.json, tools/prof_summary.py."
  python3 tools/prof_summary.py pmc $out/p1 $out/p2 $out/p3; } > $out/pmc_summary.txt
python3 tools/prof_summary.py pmcjson $out/pmc.json $out/p1 $out/p2 $out/p3 > /dev/null
rm -rf $out/p1 $out/p2 $out/p3
cat $out/pmc_summary.txt
