set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --config cfg5 --rows 2000000 --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5', d['ms_per_step'], d['value'], d['roofline']['frac'], d['kernel_ms_per_step'])"
