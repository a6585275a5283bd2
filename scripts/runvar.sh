for v in nohuman_engine nh_e1 nh_e2 nh_e3 nohuman_engine nh_e1 nh_e2 nh_e3; do
  NOHUMAN_NO_SHORT=1 NOHUMAN_ENGINE_LIB=$PWD/nohuman_amd/lib$v.so python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 30 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v','PE generic',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done
