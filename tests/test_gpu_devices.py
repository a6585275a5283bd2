"""Multi-GPU readiness that one GPU can check (configs[4] has never met an 8-GPU node).
NOHUMAN_FAKE_DEVICES=3 gives the library three LOGICAL devices on the one GPU: the run's engines, reader lanes, encoders,
streams and buffers then have three distinct owners, copies between them take the peer route (or, NOHUMAN_NO_PEER=1, a
page-locked host bounce), and NOHUMAN_DEBUG_DEVICE=1 asserts at every launch / copy / allocation site that the calling thread's
device is the owner's and that every buffer was allocated under its owner.  Child processes: the switches are read once."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "tests", "golden", "toy_db")

CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, %(root)r)
from nohuman_amd import engine
res = {}
for name, ids in (("one", [0]), ("three", [0, 1, 2])):
    for what, kw in (("plain", {}), ("gzip", dict(out_codec=2, codec_threads=2))):
        for paired in (True, False):
            o1, o2, k = (os.path.join(%(tmp)r, "%%s_%%s_%%d_%%s" %% (name, what, paired, x)) for x in ("o1", "o2", "k"))
            st = engine.run(%(db)r, %(in1)r, o1, in2=%(in2)r if paired else None, out2=o2 if paired else None, kraken_output=k,
                            device_ids=ids, threads=4, **kw)
            import gzip as gz
            rd = (lambda p: gz.decompress(open(p, "rb").read())) if what == "gzip" else (lambda p: open(p, "rb").read())
            res["%%s %%s %%d" %% (name, what, paired)] = [hashlib.sha256(rd(o1)).hexdigest(), hashlib.sha256(rd(o2)).hexdigest() if paired else "",
                                                      hashlib.sha256(open(k, "rb").read()).hexdigest(), st.total_sequences, st.classified]
print("RESULT " + json.dumps(res))
"""


def _inputs(tmp_path):
    raw1 = open(os.path.join(ROOT, "tests", "golden", "reads_pe_1.fq"), "rb").read() * 8
    raw2 = open(os.path.join(ROOT, "tests", "golden", "reads_pe_2.fq"), "rb").read() * 8
    p1, p2 = tmp_path / "r_1.fq.gz", tmp_path / "r_2.fq.gz"
    p1.write_bytes(gzip.compress(raw1, 6))
    p2.write_bytes(gzip.compress(raw2, 1))
    return str(p1), str(p2), raw1.count(b"\n") // 4


def _child(tmp_path, extra_env, code=None):
    in1, in2, n = _inputs(tmp_path)
    env = dict(os.environ, NOHUMAN_FAKE_DEVICES="3", NOHUMAN_DEBUG_DEVICE="1", NOHUMAN_RCCL="0", NOHUMAN_BATCH_FRAGS="100",
               NOHUMAN_GZDEV_SEG="16384", NOHUMAN_GZDEV_STRETCH="4096", NOHUMAN_GZDEV_MIN_BYTES="0", NOHUMAN_TRACE="1", **extra_env)
    env.pop("NOHUMAN_GZ_READER", None)
    src = (code or CHILD) % dict(root=ROOT, tmp=str(tmp_path), db=DB, in1=in1, in2=in2)
    return subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True, timeout=600), n


@pytest.mark.parametrize("no_peer", [False, True])
def test_three_logical_devices_keep_the_discipline_and_the_bytes(tmp_path, no_peer):
    import json
    out, n = _child(tmp_path, {"NOHUMAN_NO_PEER": "1"} if no_peer else {})
    assert out.returncode == 0, out.stderr[-3000:]
    assert "DEVICE DISCIPLINE" not in out.stderr, [ln for ln in out.stderr.splitlines() if "DISCIPLINE" in ln][:3]
    res = json.loads(out.stdout.split("RESULT ")[1])
    for what in ("plain", "gzip"):
        for paired in (1, 0):
            assert res["three %s %d" % (what, paired)] == res["one %s %d" % (what, paired)], (what, paired)
            assert res["one %s %d" % (what, paired)][3] == n
    # the run really spread: pieces of the gzip streams were decoded on all three devices, batches classified there
    assert "pieces by device (device:pieces)" in out.stderr
    line = [ln for ln in out.stderr.splitlines() if "pieces by device" in ln][-1]
    assert all((" %d:" % g) in line for g in (0, 1, 2)), line


def test_the_checker_catches_a_launch_under_the_wrong_device(tmp_path):
    """The same run with a knob that issues every launch under the run's first device: the discipline check must fail it."""
    out, _ = _child(tmp_path, {"NOHUMAN_DEBUG_DEVICE_BREAK": "1"})
    assert out.returncode != 0
    assert "device discipline" in out.stderr and "DEVICE DISCIPLINE" in out.stderr, out.stderr[-2000:]
