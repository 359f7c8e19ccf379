#!/bin/bash
# experiment helper: bench at several CABAC thread counts (prints value, device ms, phases)
for t in "$@"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --host-threads $t 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "import json; d=json.load(open('/tmp/b.json')); print('threads', $t, d['value'], d['host']['device_ms_per_step'], d['host']['step_phases_ms'], d['host']['entropy_ms_per_frame_sum_over_threads'])"
done
