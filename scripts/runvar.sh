for v in nh_old nohuman_engine nh_old nohuman_engine; do
  NOHUMAN_ENGINE_LIB=$PWD/nohuman_amd/lib$v.so python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 10 --ont --pairs 400000 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v','ont',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
  NOHUMAN_NO_SHORT=1 NOHUMAN_ENGINE_LIB=$PWD/nohuman_amd/lib$v.so python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 30 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v','PE generic',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done
