#!/bin/bash
# A/B: bench.py (WORKLOAD, default quads) once per library ab/lib_<name>.so given by name; run through gpurun from the repo root
for lib in "$@"; do
  v=$(PTGPU_LIB=$PWD/ab/lib_$lib.so timeout -k 10 200 python bench.py --no-cpu-baseline --workload ${WORKLOAD:-quads} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j.get('bvh',{}).get('steps_per_traced_ray'))") || exit 1
  echo "$lib $v" | tee -a gpurun_out/ab.log
done
