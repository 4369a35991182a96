#!/bin/bash
# Round profile on the GPU box: GPU tests, bench line, rocprofv3 kernel stats of the bench command, and the
# two HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).
# usage: tools/gpu_profile.sh <tag>     (outputs under gpurun_out/<tag>/)
set -o pipefail
tag=${1:-prof}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -30 "$out/tests.log"; exit 1; }
tail -2 "$out/tests.log"
timeout -k 10 300 python bench.py > "$out/bench.log" 2>&1 || { tail -30 "$out/bench.log"; exit 1; }
tail -1 "$out/bench.log"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$out/stats.log" 2>&1 || { tail -30 "$out/stats.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1 || { tail -30 "$out/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1 || { tail -30 "$out/pmc_write.log"; exit 1; }
# MFMA utilisation (row d-2): 3 SQ counters + 1 GRBM counter, a pass of its own
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_mfma.log" 2>&1 || { tail -30 "$out/pmc_mfma.log"; exit 1; }
find "$out" -name '*.csv' | head -20
du -sh "$out"
