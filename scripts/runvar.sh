for rep in 1 2; do for c in 24 31 16 12; do
NOHUMAN_FRAG_CHUNK=$c python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 40 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('chunk$c',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done; done
