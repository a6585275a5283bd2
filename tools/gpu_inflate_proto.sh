# builds the test input of tools/gpu_inflate_proto (bench.py's FASTQ text, gzip -6 by zlib) and runs the prototype;
# with tools/gpu_inflate_proto_<NEAR> present (built with -DNEAR=...), those variants too
set -e
mkdir -p /dev/shm/gip
python - <<'PY'
import sys, types, zlib, torch
sys.path.insert(0, '.')
import bench
cx = types.SimpleNamespace(torch=torch, dev=torch.device('cuda', 0))
bench.e2e_member(cx, 1_000_000, 150, 1, 0, '/dev/shm/gip/m.fq')
d = open('/dev/shm/gip/m.fq', 'rb').read()
c = zlib.compressobj(6, zlib.DEFLATED, 31)
open('/dev/shm/gip/m.fq.gz', 'wb').write(c.compress(d) + c.flush())
print(len(d), 'bytes of text')
PY
for exe in tools/gpu_inflate_proto tools/gpu_inflate_proto_*[0-9]; do
  [ -x $exe ] || continue
  echo "== $exe"
  for kib in 4096 32; do timeout 300 ./$exe /dev/shm/gip/m.fq.gz $kib || true; done
  echo "-- the whole reader: search, decode, chain, marker replacement"
  for kib in 256 64 32 16; do timeout 300 ./$exe /dev/shm/gip/m.fq.gz $kib full || true; done
done
rm -rf /dev/shm/gip
