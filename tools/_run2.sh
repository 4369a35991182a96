set -o pipefail
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 120 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['kernel_ms_per_step'])"
timeout -k 10 120 python bench.py --full --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('full', d['ms_per_step'], d['value'])"
timeout -k 10 120 python bench.py --config cfg1 --full --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg1 full', d['ms_per_step'], d['value'])"
