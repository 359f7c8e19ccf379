#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from `hipcc -Rpass-analysis=kernel-resource-usage` (stderr saved to a file):
  hipcc ... -Rpass-analysis=kernel-resource-usage -c device.hip -o /tmp/x.o 2> ra.txt ; python tools/resource_usage.py ra.txt"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split(' [')[0]
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name

    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return m.group(1) if m else '?'
    print("%-62s VGPR %4s SGPR %4s scratch %4s occ %3s LDS %6s" % (dem[:62], g('VGPRs'), g('TotalSGPRs'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]'), g('LDS Size [bytes/block]')))
