set -o pipefail
mkdir -p gpurun_out/c45e
timeout -k 10 300 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_dp.py -m gpu -x -q -k "band or pipeline or clean or world1" --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 300 python bench.py --config cfg5 --rows 10000000 --steps 1220 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/c45e/cfg5.json || exit 1
python -c "
import json; d=json.load(open('gpurun_out/c45e/cfg5.json')); print('cfg5', d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('kernel_ms_per_step'))"
