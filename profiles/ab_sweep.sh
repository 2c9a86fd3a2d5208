#!/bin/bash
# sample workload over library variants, printing the backward sweep's kernel times: bash profiles/ab_sweep.sh a.so b.so ...
show='import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms"]; print(sys.argv[1], d["value"], "shade", k["trace_shade"], "gather", k["gather"], "resolve", k["resolve"], "pipelined", d.get("pipelined",{}).get("value"))'
for rep in 1 2; do
for lib in "$@"; do
  PTGPU_LIB=$lib timeout -k 10 200 python -u bench.py --steps 16 --warmup 3 --no-cpu-baseline ${AB_ARGS} | python -c "$show" $lib || exit 1
done
done
