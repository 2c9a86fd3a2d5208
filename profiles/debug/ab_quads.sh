#!/bin/bash
# A/B of library builds on the quad field: bash profiles/debug/ab_quads.sh a.so b.so ...
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["kernel_ms"])'
for rep in 1; do
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --workload quads --steps 4 --warmup 1 --no-cpu-baseline --passes 1 | python -c "$show" $lib || exit 1
done
done
