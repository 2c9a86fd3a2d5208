#!/bin/bash
# sample workload over library variants (alternate order twice): bash profiles/ab_sample.sh a.so b.so ...
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], d["roofline"]["kernel_ms"]["trace_shade"])'
for rep in 1 2; do
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --steps 16 --warmup 3 --no-cpu-baseline | python -c "$show" $lib || exit 1
done
done
