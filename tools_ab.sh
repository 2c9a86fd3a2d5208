#!/bin/bash
# A/B on one box: alternate libraries, sample workload (and room), print Mrays/s and fused-kernel ms
for rep in 1 2; do
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --steps 32 --warmup 4 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib sample', d['value'], d['roofline']['kernel_ms'].get('trace_shade'))" || exit 1
done
done
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --workload room --steps 8 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib room', d['value'], d['roofline']['kernel_ms'].get('trace_shade'))" || exit 1
done
