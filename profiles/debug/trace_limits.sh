#!/bin/bash
# Diagnostic counter passes for the BVH trace kernel (quad field): where its waves spend their cycles and how busy the
# texture-address / L1 path is.  Run through gpurun from the repo root; results under gpurun_out/limits_*.
set -e
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${1:-quads}
SHORT="python3 bench.py --workload $WL --steps 1 --warmup 0 --spp-per-step 64 --no-cpu-baseline --no-events"
timeout -s KILL 120 rocprofv3 --list-avail > $OUT/limits_avail.txt 2>&1 || true
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/limits_sq1 -- $SHORT > $OUT/limits_sq1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH --output-format csv -d $OUT/limits_sq2 -- $SHORT > $OUT/limits_sq2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/limits_ta -- $SHORT > $OUT/limits_ta.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for tag in ("sq1", "sq2", "ta"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"gpurun_out/limits_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "trace_dyn" in k or "k_shade" in k or "k_gather" in k or "k_resolve" in k:
            print(tag, k, {a: f"{b:.4g}" for a, b in v.items()})
PY
