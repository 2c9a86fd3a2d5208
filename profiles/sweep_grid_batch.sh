#!/bin/bash
# Sample workload: blocks per tile kernel (PTGPU_GRID) x samples per pass (--batch), 8 spp per step
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"])'
for g in 4096 8192 16384; do
  for b in 2 4 8; do
    PTGPU_GRID=$g timeout -k 10 200 python -u bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-events --spp-per-step 8 --batch $b | python -c "$show" grid=$g batch=$b || exit 1
  done
done
