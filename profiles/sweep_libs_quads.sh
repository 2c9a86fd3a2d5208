#!/bin/bash
# quad-field workload over library variants: bash profiles/sweep_libs_quads.sh a.so b.so ...
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["bvh"])'
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --workload quads --steps 3 --warmup 1 --no-cpu-baseline | python -c "$show" $lib || exit 1
done
