"""Folds the rocprofv3 counter CSVs written by profiles/collect.sh into per-kernel, per-launch numbers.
The output carries `_source_hash` (bench.py's hash over the device sources the run was built from) and `_commit`
(git revision, handed in through PT_COMMIT: the GPU box has no .git), so that a reader - bench.py included - can
tell whether the figures belong to the kernels it is looking at.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-byte line requested from the memory
side, whatever part of the line the wave reads: calibrated with profiles/micro/fetch_calibration.hip
(profiles/r02_fetch_calibration.json) - a streamed 2 GiB table reads 0.500 of its bytes, and a gather of 16, 64 or
112 bytes from each of 8M distinct lines reads 64.0 B per line in all three cases.  It is doubled here, which prices
every touched line whole: exact for the streamed queue records, an upper bound for the BVH's per-lane node and leaf
fetches.  WRITE_SIZE is exact for 16-byte-per-lane stores.  Output: <out>/<tag>_counters.json and <out>/<tag>_hbm_traffic.json
(HBM bytes per launch per kernel class, the `traffic` field of bench.py's roofline block).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402

out, tag = sys.argv[1], sys.argv[2]
CLASS = [("k_shade<true", "trace_shade"), ("k_shade<false", "shade"), ("k_trace", "trace"), ("k_terminal", "terminal"),
         ("k_gather", "gather"), ("k_resolve", "resolve"), ("k_raygen", "raygen")]


def klass(name):
    for key, c in CLASS:
        if key in name:
            return c
    return None


agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(f"{out}/{tag}_pmc_*/*/*counter_collection.csv"):   # (every counter pass taken under this tag)
    for r in csv.DictReader(open(f)):
        c = klass(r["Kernel_Name"])
        if c is None:
            continue
        agg[c][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[c][r["Counter_Name"]].add(r["Dispatch_Id"])
res, traffic = {}, {}
for c, v in agg.items():
    res[c] = {}
    for name, total in v.items():
        n = max(1, len(launches[c][name]))
        res[c][name] = {"total": total, "launches": n, "per_launch": total / n}
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd = 2.0 * 1024.0 * res[c]["FETCH_SIZE"]["per_launch"]
        wr = 1024.0 * res[c]["WRITE_SIZE"]["per_launch"]
        res[c]["hbm_bytes_per_launch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr}
        traffic[c] = rd + wr
res["_source_hash"] = source_hash()
res["_commit"] = os.environ.get("PT_COMMIT", "unknown")
json.dump(res, open(f"{out}/{tag}_counters.json", "w"), indent=1)
json.dump(traffic, open(f"{out}/{tag}_hbm_traffic.json", "w"), indent=1)
print(json.dumps(traffic))
