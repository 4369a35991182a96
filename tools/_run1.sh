set -o pipefail
bash tools/gpu_profile.sh fin r03 || exit 1
bash tools/clean_profile.sh finc r03 || exit 1
bash tools/cfg_profile.sh fin4 cfg4 r03 3 > gpurun_out/fin4.log 2>&1 || { tail -20 gpurun_out/fin4.log; exit 1; }
mkdir -p gpurun_out/fin1
timeout -k 10 200 python bench.py --config cfg1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/fin1/r03_cfg1_bench.json || exit 1
timeout -k 10 200 python bench.py --full --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/fin1/r03_full_bench.json || exit 1
timeout -k 10 200 python bench.py --share-gpass --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/fin1/r03_share_bench.json || exit 1
python -c "
import json
for n in ('cfg1','full','share'):
    d=json.load(open('gpurun_out/fin1/r03_%s_bench.json'%n)); print(n, d['ms_per_step'], d['value'], d['roofline']['frac'])"
