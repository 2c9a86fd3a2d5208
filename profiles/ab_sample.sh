#!/bin/bash
# sample workload over library variants (alternate order twice): bash profiles/ab_sample.sh a.so b.so ...
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(sys.argv[1], d["value"], "one pass", d["roofline"]["measured_with"]["mrays_per_s"], {n: k[n] for n in ("trace_shade", "gather", "resolve") if n in k})'
for rep in 1 2; do
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --workload ${WL:-sample} --steps ${STEPS:-16} --warmup 3 --no-cpu-baseline | python -c "$show" $lib || exit 1
done
done
