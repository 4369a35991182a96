#!/usr/bin/env python3
"""HBM traffic per kernel from the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they cannot share a
pass on gfx950).  Units and corrections follow MI355X_MICROARCH.md (HBM section): the counters are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) coalesced reads at 64 bytes, so
the read side is doubled for kernels whose loads are 16-byte wide (all of ours); WRITE_SIZE is exact for
16-byte stores.
With a bench line (bench.py's JSON, `algo_bytes_per_launch` / `kernel_launches_per_step`) the summary also carries, per
LSTM family, counter traffic / SURVEY 8(d) algorithmic bytes, and for the whole step the counter traffic of all three
families against the step's algorithmic bytes (the GEMMs count as zero there: the model takes them as fused).
usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json] [bench.json]"""
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
                per[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    return per


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def family(n):
    for f in ("lstm_fwd_kernel", "lstm_bwd_kernel", "gemm"):
        if f in n:
            return f
    return None


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
rows, fam = [], defaultdict(lambda: [0, 0.0, 0.0])
for k in sorted(set(fetch) | set(write), key=lambda k: -(sum(fetch.get(k, [])) * 2 + sum(write.get(k, [])))):
    n = max(len(fetch.get(k, [])), len(write.get(k, [])))
    fr, wr = sum(fetch.get(k, [])), sum(write.get(k, []))
    rows.append(dict(kernel=short(k), launches=n, fetch_raw_bytes_per_launch=fr / n, fetch_corrected_bytes_per_launch=2 * fr / n,
                     write_bytes_per_launch=wr / n, hbm_bytes_per_launch=(2 * fr + wr) / n))
    f = family(k)
    if f:
        fam[f][0] += n
        fam[f][1] += 2 * fr
        fam[f][2] += wr
out = dict(note="FETCH_SIZE doubled (gfx950 wide-read correction), KiB -> bytes; separate PMC passes of bench.py --steps 2 --warmup 1",
           families={f: dict(launches=v[0], hbm_bytes_per_launch=(v[1] + v[2]) / max(1, v[0]),
                             read_bytes_per_launch=v[1] / max(1, v[0]), write_bytes_per_launch=v[2] / max(1, v[0]))
                     for f, v in fam.items()},
           kernels=rows)
if len(sys.argv) > 4:
    with open(sys.argv[4]) as fb:
        bench = json.loads([l for l in fb.read().splitlines() if l.startswith("{")][-1])
    algo = bench.get("algo_bytes_per_launch", {})
    lps = bench.get("kernel_launches_per_step", {})
    step_traffic = 0.0
    for f, key in (("lstm_fwd_kernel", "lstm_fwd"), ("lstm_bwd_kernel", "lstm_bwd"), ("gemm", "gemm")):
        if f in out["families"]:
            v = out["families"][f]
            if key in algo:
                v["algo_bytes_per_launch"] = algo[key]
                v["traffic_over_algorithmic"] = v["hbm_bytes_per_launch"] / algo[key]
            if key in lps:
                v["launches_per_step"] = lps[key]
                step_traffic += v["hbm_bytes_per_launch"] * lps[key]
    abw = bench.get("step_hbm_roofline", {}).get("algo_bytes_per_window")
    gb = bench.get("config", {}).get("global_batch")
    if abw and gb and step_traffic:
        out["step"] = dict(counter_bytes_per_step=step_traffic, algo_bytes_per_step=abw * gb / max(1, bench.get("n_gpus", 1)),
                           traffic_over_algorithmic=step_traffic / (abw * gb / max(1, bench.get("n_gpus", 1))),
                           note="SURVEY 8(d)'s per-window figure leaves the critic step out (forecast_len windows per iteration: "
                                "128 beside a batch of 512 at cfg2; 1024 beside a batch of 64 at cfg4, where this ratio "
                                "therefore says nothing -- use the per-family ratios)")
        print(f"step: {step_traffic / 1e9:.2f} GB counter traffic vs {out['step']['algo_bytes_per_step'] / 1e9:.2f} GB algorithmic "
              f"= {out['step']['traffic_over_algorithmic']:.2f}x")
for r in rows[:25]:
    print(f"{r['kernel'][:60]:60s} n={r['launches']:4d} read={r['fetch_corrected_bytes_per_launch'] / 1e6:9.2f} MB write={r['write_bytes_per_launch'] / 1e6:9.2f} MB")
for f, v in out["families"].items():
    extra = f", {v['traffic_over_algorithmic']:.2f}x algorithmic" if "traffic_over_algorithmic" in v else ""
    print(f"family {f}: {v['launches']} launches, {v['hbm_bytes_per_launch'] / 1e6:.2f} MB/launch{extra}")
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fo:
        json.dump(out, fo, indent=1)
