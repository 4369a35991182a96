#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 PMC pass of bench.py (north_star: "rocprof counters must show MFMA
utilisation on the projection GEMMs").

Counters of the pass (tools/gpu_profile.sh): SQ_INSTS_VALU_MFMA_MOPS_BF16 (bf16 matrix work in units of 512
flops -- every product here is three v_mfma_f32_16x16x32_bf16, the "bf16x3" split, so the useful fp32-grade
flops are a third of it), SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES (both summed over the chip by rocprofv3),
GRBM_GUI_ACTIVE (kernel cycles).  Reported per kernel name (one per template instantiation):
  bf16_tflops   = MOPS * 512 / duration            (dense bf16 MFMA issue, peak 2.5 PFLOP/s)
  mfma_frac     = bf16_tflops / 2500
  mfma_busy     = MFMA_BUSY_CYCLES / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)   (rocprof's MfmaUtil; rocprofv3 reports
                  GRBM_GUI_ACTIVE summed over the 8 XCDs, hence the / 8)
Durations are the dispatch timestamps of the same pass (counter collection serialises kernels).
usage: python tools/mfma_summary.py <counter_collection.csv> [out.json]"""
import csv
import json
import re
import sys
from collections import defaultdict

PEAK_BF16_TFLOPS = 2500.0


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


disp = {}
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        d = disp.setdefault(r["Dispatch_Id"], dict(name=short(r["Kernel_Name"]), c={},
                                                     dur=(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9))
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
per = defaultdict(lambda: dict(launches=0, seconds=0.0, mops=0.0, mfma_busy=0.0, sq_busy=0.0, gui=0.0))
for d in disp.values():
    p = per[d["name"]]
    p["launches"] += 1
    p["seconds"] += d["dur"]
    p["mops"] += d["c"].get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    p["mfma_busy"] += d["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    p["sq_busy"] += d["c"].get("SQ_BUSY_CYCLES", 0.0)
    p["gui"] += d["c"].get("GRBM_GUI_ACTIVE", 0.0)
rows = []
for name, p in sorted(per.items(), key=lambda kv: -kv[1]["mops"]):
    if p["mops"] <= 0:
        continue
    tf = p["mops"] * 512 / p["seconds"] / 1e12 if p["seconds"] > 0 else 0.0
    rows.append(dict(kernel=name, launches=p["launches"], avg_us=p["seconds"] / p["launches"] * 1e6,
                     bf16_tflops=tf, mfma_frac=tf / PEAK_BF16_TFLOPS, fp32_equiv_tflops=tf / 3,
                     mfma_busy=(p["mfma_busy"] / (p["gui"] / 8 * 256 * 4)) if p["gui"] > 0 else None,
                     mfma_busy_of_sq_busy=(p["mfma_busy"] / p["sq_busy"]) if p["sq_busy"] > 0 else None))
fam = {}
for key in ("gemm", "lstm_fwd_kernel", "lstm_bwd_kernel"):
    sel = [r for r in rows if key in r["kernel"]]
    if sel:
        sec = sum(r["avg_us"] * r["launches"] for r in sel) * 1e-6
        fl = sum(r["bf16_tflops"] * r["avg_us"] * r["launches"] for r in sel) * 1e-6
        fam[key] = dict(launches=sum(r["launches"] for r in sel), bf16_tflops=fl / sec if sec else 0.0,
                        mfma_frac=fl / sec / PEAK_BF16_TFLOPS if sec else 0.0, seconds=sec)
out = dict(note="rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                "--kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline; bf16 MFMA flops = MOPS * 512; "
                "every fp32-grade product is three bf16 MFMAs (useful flops = a third); peak 2.5 PFLOP/s dense bf16",
           peak_bf16_tflops=PEAK_BF16_TFLOPS, families=fam, kernels=rows)
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fo:
        json.dump(out, fo, indent=1)
for r in rows[:20]:
    busy = "n/a" if r["mfma_busy"] is None else f"{100 * r['mfma_busy']:.1f}%"
    print(f"{r['kernel'][:70]:70s} n={r['launches']:4d} {r['avg_us']:9.1f} us  {r['bf16_tflops']:8.1f} TF bf16 "
          f"({100 * r['mfma_frac']:.1f}% of peak)  busy {busy}")
