#!/bin/bash
# Round profile on the GPU box: GPU tests, bench line, rocprofv3 kernel stats of the bench command, the two HBM-traffic
# PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950) and the MFMA pass, then the summaries.
# usage: tools/gpu_profile.sh <tag> [round]    (outputs under gpurun_out/<tag>/; summaries named <round>_*)
set -o pipefail
tag=${1:-prof}
rnd=${2:-r03}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$out/tests.log" 2>&1 || { tail -30 "$out/tests.log"; exit 1; }
tail -2 "$out/tests.log"
timeout -k 10 300 python bench.py > "$out/bench.log" 2>&1 || { tail -30 "$out/bench.log"; exit 1; }
tail -1 "$out/bench.log" > "$out/${rnd}_bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$out/stats.log" 2>&1 || { tail -30 "$out/stats.log"; exit 1; }
grep "^{\"metric\"" "$out/stats.log" | tail -1 > "$out/${rnd}_bench_profiled.json"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/${rnd}_kernel_stats.csv"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1 || { tail -30 "$out/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1 || { tail -30 "$out/pmc_write.log"; exit 1; }
# MFMA utilisation (row d-2): 3 SQ counters + 1 GRBM counter, a pass of its own
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_mfma.log" 2>&1 || { tail -30 "$out/pmc_mfma.log"; exit 1; }
python tools/pmc_summary.py "$(find "$out/pmc_fetch" -name '*counter_collection.csv' | head -1)" "$(find "$out/pmc_write" -name '*counter_collection.csv' | head -1)" "$out/${rnd}_traffic.json" "$out/${rnd}_bench.json" | tail -12
python tools/mfma_summary.py "$(find "$out/pmc_mfma" -name '*counter_collection.csv' | head -1)" "$out/${rnd}_mfma.json" > "$out/mfma.txt"; sed -n 1,12p "$out/mfma.txt"
rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_mfma"
tail -1 "$out/bench.log"
