#!/bin/bash
# usage (on the GPU box): bash tools_quadsweep.sh lib1.so lib2.so ...   -> quads bench per library variant
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --workload quads --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']['trace'])" || exit 1
done
