#!/bin/bash
# FETCH_SIZE against known byte counts (profiles/micro/fetch_calibration.hip); run on the GPU box from the repo root.
set -e
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/fetch_calibration profiles/micro/fetch_calibration.hip
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- $OUT/fetch_calibration > $OUT/cal_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/cal_rdreq -- $OUT/fetch_calibration > $OUT/cal_rdreq.log 2>&1 || true
python3 - <<'PY'
import csv, glob, json
res = {}
for sub in ("cal_fetch", "cal_rdreq"):
    for f in glob.glob(f"gpurun_out/{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            res.setdefault(name, {})[r["Counter_Name"]] = res.get(name, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
lines = (1 << 24) // 2
expect = {"k_stream": (1 << 24) * 128}
out = {}
for name, c in sorted(res.items()):
    if "FETCH_SIZE" not in c:
        continue
    got = c["FETCH_SIZE"] * 1024.0
    row = {"FETCH_SIZE_bytes": got, **{k: v for k, v in c.items() if k != "FETCH_SIZE"}}
    if "k_stream" in name:
        row["known_bytes"] = expect["k_stream"]; row["counter_over_known"] = got / expect["k_stream"]
    elif "k_gather" in name:
        row["lines_touched"] = lines; row["counter_bytes_per_line"] = got / lines
    out[name] = row
json.dump(out, open("gpurun_out/fetch_calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
