#!/bin/bash
# A/B of library builds on the two BVH workloads: bash profiles/debug/ab_bvh.sh a.so b.so ...
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["kernel_ms"])'
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --workload room --accel 2 --steps 4 --warmup 1 --no-cpu-baseline | python -c "$show" $lib room-bvh || exit 1
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --workload quads --steps 4 --warmup 1 --no-cpu-baseline | python -c "$show" $lib quads || exit 1
done
