#!/bin/bash
# cfg5 (cleaning pass) profile: bench line of the full 10M-row sweep, rocprofv3 kernel stats and the two HBM-traffic passes
# of a shorter run.  usage: tools/clean_profile.sh <tag> [round]
set -o pipefail
tag=${1:-cprof}; rnd=${2:-r03}
out=$PWD/gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
ARGS="--config cfg5 --rows 1000000 --steps 20 --warmup 2 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py $ARGS > "$out/stats.log" 2>&1 || { tail -20 "$out/stats.log"; exit 1; }
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/${rnd}_clean_kernel_stats.csv"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pf" -- python3 bench.py $ARGS > "$out/pf.log" 2>&1 || { tail -20 "$out/pf.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pw" -- python3 bench.py $ARGS > "$out/pw.log" 2>&1 || { tail -20 "$out/pw.log"; exit 1; }
# steps run by that command: warm-up 2 + timed 20 + 5 with the event timers
python tools/clean_traffic_summary.py "$(find "$out/pf" -name '*counter_collection.csv' | head -1)" "$(find "$out/pw" -name '*counter_collection.csv' | head -1)" 27 "$out/${rnd}_clean_traffic.json"
rm -rf "$out/stats" "$out/pf" "$out/pw"
mkdir -p profiles && cp "$out/${rnd}_clean_traffic.json" profiles/     # the sweep's bench line reads it (roofline.traffic)
timeout -k 10 300 python bench.py --config cfg5 --rows 10000000 --steps 1220 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > "$out/${rnd}_cfg5_10m_bench.json"
cat "$out/${rnd}_cfg5_10m_bench.json"
