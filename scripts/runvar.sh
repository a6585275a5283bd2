for rep in 1 2; do for c in 2 4 8; do
NOHUMAN_TABLE_COPIES=$c python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 40 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('copies$c',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done; done
