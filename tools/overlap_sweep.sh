#!/bin/bash
# bench line for each TB_OVERLAP mask (1 = GF under critic, 2 = GF2 under D backward, 4 = graph cut)
set -o pipefail
out=gpurun_out/${1:-ov}; mkdir -p $out
for m in 5 7 1 4 0; do
  TB_OVERLAP=$m timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $out/ov$m.json || exit 1
  python - $m $out/ov$m.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print("mask", sys.argv[1], d["value"], "windows/s", d["ms_per_step"], "ms", d["kernel_ms_per_step"], flush=True)
PY
done
