#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd database (the same columns as --stats' kernel_stats.csv).
usage: python tools/rocpd_stats.py results.db [out.csv] [--per N]   (--per N: also print ms per N-th of the calls)"""
import csv
import re
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
per = int(sys.argv[sys.argv.index("--per") + 1]) if "--per" in sys.argv else 0
if "--per" in sys.argv:
    args = [a for a in args if a != str(per)]
db = sqlite3.connect(args[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = {}
for name, start, end in db.execute(f"select {name_col}, start, end from kernels"):
    rows.setdefault(name, []).append(end - start)
total = sum(sum(v) for v in rows.values())
table = []
for name, v in rows.items():
    n = len(v)
    mean = sum(v) / n
    sd = (sum((x - mean) ** 2 for x in v) / n) ** 0.5
    table.append((name, n, sum(v), mean, 100.0 * sum(v) / total, min(v), max(v), sd))
table.sort(key=lambda r: -r[2])
if len(args) > 1:
    with open(args[1], "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        w.writerows(table)
for r in table[:40]:
    short = re.sub(r"\(anonymous namespace\)::", "", r[0])
    short = re.sub(r"\(.*", "", short)[:70]
    extra = f"  {r[2] / per / 1e6:7.3f} ms/iter" if per else ""
    print(f"{short:70s} n={r[1]:5d} avg={r[3] / 1e3:9.1f}us tot={r[2] / 1e6:8.2f}ms {r[4]:5.1f}%{extra}")
print(f"total kernel time {total / 1e6:.2f} ms" + (f" = {total / per / 1e6:.3f} ms/iter" if per else ""))
