set -o pipefail
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
for i in 1 2; do timeout -k 10 120 python bench.py --config cfg1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg1', d['ms_per_step'], d['value'], d['kernel_ms_per_step'])"; done
