for mode in "--ont --pairs 400000 --steps 10" "--steps 30" ; do
  NOHUMAN_NO_SHORT=1 python bench.py --no-e2e --no-variants --no-cpu-baseline $mode | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('generic','$mode',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done
python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 20 --n-rate 0.001 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('short n-rate 0.001',d['value'],d['roofline']['frac'])"
python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 20 --hit-frac 1.0 --pairs 1000000 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('short hit 1.0',d['value'],d['roofline']['frac'],d['config']['classified_fraction'])"
python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 20 --hit-frac 1.0 --pairs 1000000 --confidence 0.5 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('short hit 1.0 conf .5',d['value'],d['roofline']['frac'],d['config']['classified_fraction'])"
