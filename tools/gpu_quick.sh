#!/bin/bash
# quick GPU iteration: tests, micro benchmarks, bench line.   usage: tools/gpu_quick.sh <tag> [micro...]
set -o pipefail
tag=${1:-q}; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -1 $out/tests.log
for m in "$@"; do timeout -k 10 200 python tools/${m}_micro.py 2>&1 | grep -v amdgpu.ids | tee $out/$m.log || exit 1; done
timeout -k 10 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $out/bench.json
