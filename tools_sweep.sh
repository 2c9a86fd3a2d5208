#!/bin/bash
# tuning helper: bench.py under several PTGPU_GRID values (run on the GPU box)
for g in "$@"; do
  echo -n "grid $g: "
  PTGPU_GRID=$g timeout -k 10 120 python -u bench.py --no-cpu-baseline --fused 1 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"])'
done
