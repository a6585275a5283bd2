for rep in 1 2; do for v in nohuman_engine nh_fs; do
  NOHUMAN_ENGINE_LIB=$PWD/nohuman_amd/lib$v.so python bench.py --no-e2e --no-variants --no-cpu-baseline --steps 40 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['roofline']['kernel_ms'],d['config']['lookups_per_read'])"
done; done
