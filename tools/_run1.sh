set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_pipeline.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -8 || exit 1
for g in "" "--gathered"; do timeout -k 10 300 python bench.py --config cfg5 --rows 2000000 --steps 100 --warmup 5 --no-cpu-baseline $g 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 $g', d['ms_per_step'], d['value'], d['roofline']['frac'], d['kernel_ms_per_step'])" || exit 1; done
