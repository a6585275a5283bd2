for rep in 1 2 3; do for v in nh_old nohuman_engine; do
NOHUMAN_ENGINE_LIB=$PWD/nohuman_amd/lib$v.so python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 40 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v PE 2.5M',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
NOHUMAN_ENGINE_LIB=$PWD/nohuman_amd/lib$v.so python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 40 --single-end --pairs 1000000 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v SE 1M',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done; done
