#!/usr/bin/env python3
"""Timeline of the LAST step in a rocprofv3 kernel_trace.csv: one line per dispatch with its start
offset, duration, gap to the previous kernel and grid -- to see per-shape GEMM times and idle gaps.
usage: python tools/trace_timeline.py kernel_trace.csv [nsteps] [--agg] [--step=N]   (N: 0-based step of the process)"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 6
# a step starts at each window_gather_kernel that follows an adam kernel; simpler: split at window_gather with grid of the big batch
idx = [i for i, r in enumerate(rows) if "window_gather" in r["Kernel_Name"]]
# two gathers per step (critic batch + main batch): steps begin at every second gather
starts = idx[::2]
which = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--step=")]
if which:
    lo, hi = starts[which[0]], starts[which[0] + 1]
else:
    lo = starts[-2] if len(starts) >= 2 else 0
    hi = starts[-1] if len(starts) >= 2 else len(rows)
seg = rows[lo:hi]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return n[:58]


agg = defaultdict(lambda: [0, 0])
busy = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    key = (short(r["Kernel_Name"]), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    agg[key][0] += 1
    agg[key][1] += e - s
    busy += e - s
    if "--agg" not in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f}us +{(e - s) / 1e3:8.1f}us gap{(s - prev_end) / 1e3:7.1f} q{r.get('Queue_Id', '?')} {key[0]} grid={key[1]},{key[2]},{key[3]}")
    prev_end = max(prev_end, e)
span = int(seg[-1]["End_Timestamp"]) - t0
print(f"step span {span / 1e6:.3f} ms, kernel busy {busy / 1e6:.3f} ms, {len(seg)} dispatches")
small = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 12000]
print(f"dispatches under 12 us: {len(small)}, {sum(small) / 1e6:.3f} ms in total")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[1] / 1e3:9.1f}us n={v[0]:3d} avg={v[1] / v[0] / 1e3:8.1f}us {k[0]} grid={k[1]},{k[2]},{k[3]}")
# concurrency of the last step: union of busy intervals vs their sum, and per-queue busy time
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
union, cur_s, cur_e = 0, None, None
for s_, e_ in iv:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
union += (cur_e - cur_s) if cur_e is not None else 0
print(f"union busy {union / 1e6:.3f} ms vs sum {busy / 1e6:.3f} ms (overlap {100 * (1 - union / busy):.1f} %), idle {(span - union) / 1e6:.3f} ms")
perq = defaultdict(int)
for r in seg:
    perq[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("busy per queue (ms):", {k: round(v / 1e6, 3) for k, v in perq.items()})
