#!/bin/bash
# A/B of library builds on one box (box-to-box variance is several per cent): bash profiles/ab.sh old.so new.so
# alternates the libraries through PTGPU_LIB: sample workload twice, room once; prints Mrays/s and per-kernel ms
show='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], d["value"], d["roofline"]["kernel_ms"])'
for rep in 1 2; do
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --steps 32 --warmup 4 --no-cpu-baseline | python -c "$show" $lib sample || exit 1
done
done
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --workload room --steps 8 --warmup 2 --no-cpu-baseline | python -c "$show" $lib room || exit 1
done
