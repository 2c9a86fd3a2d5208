#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default run on the GPU box (run through gpurun from the repo
# root).  Kernel times and hardware counters come from SEPARATE runs (counters never together with tracing).
# The counter runs come first and their summary is put where bench.py looks for it, so that the line printed by the traced
# run carries traffic and instruction figures of the same sources.  The traced run is the leg of bench.py's default
# command that supplies its roofline block: ONE pass in flight (--passes 1) of the pass size the library chooses by
# default (--batch, asked of the library first), without the frame loop.  Kernels of two passes in flight overlap and each
# reports the time it shared the GPU; with one pass the kernel statistics cover the same launches the HIP events of that
# leg time, and the two averages can be compared.
# Usage: PT_COMMIT=<git rev> bash profiles/collect.sh <tag> [workload]   -> gpurun_out/<tag>_*   (copy what should be kept into profiles/)
set -e
TAG=${1:-r03}
WL=${2:-sample}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEPS=4; SSTEPS=1
B=$(python3 bench.py --workload $WL --steps 1 --warmup 0 --spp-per-step 1 --no-cpu-baseline --no-events --frame-loop 0 | python3 -c 'import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["config"]["samples_per_pass"])')
echo "samples per pass (library default for $WL): $B"
SHORT="python3 bench.py --workload $WL --steps $SSTEPS --warmup 0 --spp-per-step $((B * 6)) --passes 1 --batch $B --no-cpu-baseline --no-events --frame-loop 0"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $SHORT > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- $SHORT > $OUT/${TAG}_pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/${TAG}_pmc_sq -- $SHORT > $OUT/${TAG}_pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/${TAG}_pmc_tcc -- $SHORT > $OUT/${TAG}_pmc_tcc.log 2>&1
python3 profiles/summarize.py $OUT $TAG
cp $OUT/${TAG}_counters.json profiles/${TAG}_counters.json   # (the box's copy of the tree: bench.py reads it from there)
BENCH="python3 bench.py --workload $WL --steps $STEPS --warmup 1 --passes 1 --batch $B --no-cpu-baseline --frame-loop 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $BENCH > $OUT/${TAG}_trace.log 2>&1
cp "$(find $OUT/${TAG}_trace -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_kernel_stats.csv
grep '^{' $OUT/${TAG}_trace.log > $OUT/${TAG}_bench_under_rocprof.json || true
