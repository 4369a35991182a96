set -o pipefail
mkdir -p gpurun_out/t5
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 | tee gpurun_out/t5/tests.log || exit 1
TB_GEMM_BIG=1 timeout -k 10 200 python tools/micro/wgrad_micro.py 2>&1 | grep "B=" || exit 1
timeout -k 10 120 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/t5/bench.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['single_stream']['frac'], d['kernel_ms_per_step'])"
