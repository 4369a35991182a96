#!/usr/bin/env python3
"""HBM traffic of ONE cleaning launch (bench.py --config cfg5) from the two rocprofv3 PMC passes, summed over the kernels
of the step (those launched a whole number of times per step; dataset set-up kernels run once and drop out).  Units and
the gfx950 wide-read correction as in tools/pmc_summary.py.
usage: python tools/clean_traffic_summary.py <fetch.csv> <write.csv> <steps_run> [out.json]"""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path, counter):
    per = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                per[re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]))].append(float(r["Counter_Value"]) * 1024.0)
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
rows, total = [], 0.0
for k in sorted(set(fetch) | set(write)):
    n = max(len(fetch.get(k, [])), len(write.get(k, [])))
    if n < steps or n % steps:
        continue
    rd, wr = 2 * sum(fetch.get(k, [])) / steps, sum(write.get(k, [])) / steps
    rows.append(dict(kernel=k, launches_per_step=n // steps, read_bytes_per_step=rd, write_bytes_per_step=wr))
    total += rd + wr
rows.sort(key=lambda r: -(r["read_bytes_per_step"] + r["write_bytes_per_step"]))
out = dict(note="FETCH_SIZE doubled (gfx950 wide-read correction), KiB -> bytes; separate PMC passes of bench.py --config cfg5; "
                "per cleaning launch (one step of 8192 windows), all kernels of the step",
           hbm_bytes_per_step=total, kernels=rows)
for r in rows:
    print(f"{r['kernel'][:64]:64s} x{r['launches_per_step']:2d} read {r['read_bytes_per_step'] / 1e6:9.1f} MB write {r['write_bytes_per_step'] / 1e6:9.1f} MB")
print(f"total {total / 1e9:.3f} GB per launch")
if len(sys.argv) > 4:
    with open(sys.argv[4], "w") as fo:
        json.dump(out, fo, indent=1)
