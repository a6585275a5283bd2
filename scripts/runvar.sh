for v in nohuman_engine nh_f nh_g nh_h nohuman_engine nh_f nh_g nh_h nohuman_engine nh_f nh_g nh_h; do
  NOHUMAN_TABLE_COPIES=2 NOHUMAN_ENGINE_LIB=$PWD/nohuman_amd/lib$v.so python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 40 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
done
