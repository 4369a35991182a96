set -o pipefail
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/tl1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/tr -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline > $out/log.txt 2>&1 || { tail -20 $out/log.txt; exit 1; }
f=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python tools/trace_timeline.py $f > $out/timeline.txt
tail -60 $out/timeline.txt
rm -rf $out/tr
