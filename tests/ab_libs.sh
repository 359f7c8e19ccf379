#!/bin/bash
# A/B several builds of libmihevc inside ONE gpurun call (boxes differ by ~10 % between calls): bash tests/ab_libs.sh <lib.so>... [-- bench args]
libs=(); args=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; args=("$@"); break; fi; libs+=("$1"); shift; done
for l in "${libs[@]}"; do
  MIHEVC_LIBRARY=$PWD/$l python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "${args[@]}" > gpurun_out/ab_tmp.json || exit 1
  python3 - "$l" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "fps  device", d["host"]["device_ms_per_step"], "ms ", d["quality"], {k: v for k, v in d["stages_ms_per_picture"].items()})
PY
done
