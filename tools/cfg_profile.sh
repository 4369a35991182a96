#!/bin/bash
# Kernel stats + HBM-traffic passes of bench.py for one of the non-headline training configs (cfg1, cfg4).
# usage: tools/cfg_profile.sh <tag> <config> [round] [steps]   (outputs under gpurun_out/<tag>/, named <round>_<config>_*)
set -o pipefail
tag=${1:-cprof}
cfg=${2:-cfg4}
rnd=${3:-r03}
steps=${4:-5}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline > "$out/bench.log" 2>&1 || { tail -30 "$out/bench.log"; exit 1; }
tail -1 "$out/bench.log" > "$out/${rnd}_${cfg}_bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline > "$out/stats.log" 2>&1 || { tail -30 "$out/stats.log"; exit 1; }
grep "^{\"metric\"" "$out/stats.log" | tail -1 > "$out/${rnd}_${cfg}_bench_profiled.json"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/${rnd}_${cfg}_kernel_stats.csv"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1 || { tail -30 "$out/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1 || { tail -30 "$out/pmc_write.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_mfma.log" 2>&1 || { tail -30 "$out/pmc_mfma.log"; exit 1; }
python tools/pmc_summary.py "$(find "$out/pmc_fetch" -name '*counter_collection.csv' | head -1)" "$(find "$out/pmc_write" -name '*counter_collection.csv' | head -1)" "$out/${rnd}_${cfg}_traffic.json" "$out/${rnd}_${cfg}_bench.json" | tail -14
python tools/mfma_summary.py "$(find "$out/pmc_mfma" -name '*counter_collection.csv' | head -1)" "$out/${rnd}_${cfg}_mfma.json" > "$out/mfma.txt"; sed -n 1,14p "$out/mfma.txt"
rm -rf "$out/stats" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_mfma"
sed -n 1,25p "$out/${rnd}_${cfg}_kernel_stats.csv" | cut -c1-220
