set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_pipeline.py -m gpu -x -q -k "dimband or band or cfg5 or clean" --timeout 200 --timeout-method thread 2>&1 | tail -15 || exit 1
mkdir -p gpurun_out/c45f
timeout -k 10 300 python bench.py --config cfg5 --rows 10000000 --steps 1220 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/c45f/cfg5.json || exit 1
python -c "
import json; d=json.load(open('gpurun_out/c45f/cfg5.json')); print('cfg5', d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('kernel_ms_per_step'))"
