#!/bin/bash
# Memory-side counters of the backward sweep (k_gather, k_resolve) on the sample scene: request sizes at the HBM interface,
# stalls of the vector cache and of L2.  One rocprofv3 run per counter group (counters never together with tracing).
# Usage (GPU box, repo root): bash profiles/debug/sweep_counters.sh [tag]   -> gpurun_out/<tag>_counters.json
set -e
TAG=${1:-sweep}
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B=$(python3 bench.py --steps 1 --warmup 0 --spp-per-step 1 --no-cpu-baseline --no-events --frame-loop 0 | python3 -c 'import sys, json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["config"]["samples_per_pass"])')
SHORT="python3 bench.py --steps 1 --warmup 0 --spp-per-step $((B * 6)) --passes 1 --batch $B --no-cpu-baseline --no-events --frame-loop 0"
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/${TAG}_pmc_ea -- $SHORT > $OUT/${TAG}_pmc_ea.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_LEVEL_sum --output-format csv -d $OUT/${TAG}_pmc_l2 -- $SHORT > $OUT/${TAG}_pmc_l2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_l1 -- $SHORT > $OUT/${TAG}_pmc_l1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $SHORT > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- $SHORT > $OUT/${TAG}_pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_sq -- $SHORT > $OUT/${TAG}_pmc_sq.log 2>&1
python3 profiles/summarize.py $OUT $TAG
