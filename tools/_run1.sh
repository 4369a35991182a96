set -o pipefail
mkdir -p gpurun_out/t7
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 | tee gpurun_out/t7/tests.log || exit 1
timeout -k 10 300 python bench.py --config cfg4 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/t7/r03_cfg4_bench.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg4', d['ms_per_step'], d['value'], d['roofline']['frac'], d['kernel_ms_per_step'])"
timeout -k 10 120 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['kernel_ms_per_step'])"
