for v in nh_old nohuman_engine nh_old nohuman_engine nh_old nohuman_engine; do
  NOHUMAN_TABLE_COPIES=2 NOHUMAN_ENGINE_LIB=$PWD/nohuman_amd/lib$v.so python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 40 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done
for i in 1 2; do NOHUMAN_NO_SHORT=1 NOHUMAN_TABLE_COPIES=2 python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 40 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('new generic',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"; done
