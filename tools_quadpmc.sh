#!/bin/bash
# TA/TCP counters for the quad-field trace kernel (diagnostic)
set -e
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SHORT="python3 bench.py --workload quads --steps 2 --warmup 1 --no-cpu-baseline --no-events"
timeout -s KILL 60 rocprofv3 -L > $OUT/avail.txt 2>&1 || true
timeout -s KILL 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/q_pmc_ta -- $SHORT > $OUT/q_pmc_ta.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/q_pmc_ta2 -- $SHORT > $OUT/q_pmc_ta2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("q_pmc_ta", "q_pmc_ta2"):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "trace" in k: print(d, k, dict(v))
PY
