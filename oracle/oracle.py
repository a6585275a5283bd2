"""ctypes binding of oracle/libk2oracle.so (k2_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

AMBIG = 0xFFFFFFFF
BORDER = 0xFFFFFFFE

RESULT_DTYPE = np.dtype([("call", "<u4"), ("total_kmers", "<u4"), ("clade_hits", "<u4"),
                         ("hit_groups", "<u4")])


class _Opts(C.Structure):
    _fields_ = [("k", C.c_uint64), ("l", C.c_uint64), ("spaced_seed_mask", C.c_uint64),
                ("toggle_mask", C.c_uint64), ("dna_db", C.c_uint64),
                ("minimum_acceptable_hash_value", C.c_uint64), ("revcom_version", C.c_int32),
                ("db_version", C.c_int32), ("db_type", C.c_int32)]


class _DB(C.Structure):
    _fields_ = [("opts", _Opts), ("capacity", C.c_uint64), ("size", C.c_uint64),
                ("key_bits", C.c_uint64), ("value_bits", C.c_uint64), ("cells", C.c_void_p),
                ("node_count", C.c_uint64), ("parent", C.POINTER(C.c_uint32)),
                ("external_id", C.POINTER(C.c_uint64)), ("linear_probing", C.c_int),
                ("reset_per_mate", C.c_int), ("minimum_hit_groups", C.c_uint32),
                ("ambiguity_rule", C.c_int), ("own_cells", C.c_void_p)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libk2oracle.so")
    src = os.path.join(_HERE, "k2_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libk2oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build() if os.path.exists(os.path.join(_HERE, "k2_oracle.c")) and _have_cc() \
            else os.path.join(_HERE, "libk2oracle.so")
        # K2ORACLE_LIB: another build of the same source (the sanitizer build of tests/test_sanitizers.py)
        so = os.environ.get("K2ORACLE_LIB") or so
        L = C.CDLL(so)
        L.k2o_last_error.restype = C.c_char_p
        L.k2o_fmix64.restype = C.c_uint64
        L.k2o_fmix64.argtypes = [C.c_uint64]
        L.k2o_reverse_complement.restype = C.c_uint64
        L.k2o_reverse_complement.argtypes = [C.c_uint64, C.c_uint, C.c_int]
        L.k2o_table_get.restype = C.c_uint32
        L.k2o_table_get.argtypes = [C.POINTER(_DB), C.c_uint64]
        L.k2o_scan_minimizers.restype = C.c_size_t
        L.k2o_scan_minimizers.argtypes = [C.POINTER(_DB), C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.c_void_p, C.c_size_t]
        L.k2o_db_from_images.argtypes = [C.POINTER(_DB), C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_size_t]
        L.k2o_db_load_dir.argtypes = [C.POINTER(_DB), C.c_char_p]
        L.k2o_db_free.argtypes = [C.POINTER(_DB)]
        L.k2o_classify.argtypes = [C.POINTER(_DB), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                   C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_uint64]
        L.k2o_classify_mt.argtypes = [C.POINTER(_DB), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                                      C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _have_cc() -> bool:
    from shutil import which
    return which("gcc") is not None and which("make") is not None


class OracleDB:
    """A kraken2 database held on the host for the CPU oracle."""

    def __init__(self, opts: bytes = None, taxo: bytes = None, hashb=None, *, directory=None,
                 cells=None, header=None):
        self._L = lib()
        self._db = _DB()
        self._keep = []
        if directory is not None:
            rc = self._L.k2o_db_load_dir(C.byref(self._db), os.fsencode(directory))
        else:
            if cells is not None:
                # cells: numpy u32 array (possibly huge); header: (capacity,size,key_bits,value_bits)
                hb = np.empty(8 + cells.size, dtype=np.uint32)
                hb[:8] = np.array(header, dtype=np.uint64).view(np.uint32)
                hb[8:] = cells
                hashb = hb
            if isinstance(hashb, (bytes, bytearray)):
                hashb = np.frombuffer(hashb, dtype=np.uint8)
            hashb = np.ascontiguousarray(hashb)
            self._keep = [opts, taxo, hashb]
            rc = self._L.k2o_db_from_images(C.byref(self._db), opts, len(opts), taxo, len(taxo),
                                            hashb.ctypes.data, hashb.nbytes)
        if rc != 0:
            raise RuntimeError(self._L.k2o_last_error().decode())

    def close(self):
        if self._db is not None:
            self._L.k2o_db_free(C.byref(self._db))
            self._db = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # switches
    def set(self, *, linear_probing=None, reset_per_mate=None, minimum_hit_groups=None, ambiguity_rule=None):
        if ambiguity_rule is not None:
            self._db.ambiguity_rule = int(ambiguity_rule)
        if linear_probing is not None:
            self._db.linear_probing = int(linear_probing)
        if reset_per_mate is not None:
            self._db.reset_per_mate = int(reset_per_mate)
        if minimum_hit_groups is not None:
            self._db.minimum_hit_groups = int(minimum_hit_groups)

    @property
    def k(self):
        return int(self._db.opts.k)

    @property
    def l(self):
        return int(self._db.opts.l)

    @property
    def node_count(self):
        return int(self._db.node_count)

    @property
    def external_ids(self):
        return np.array([self._db.external_id[i] for i in range(self.node_count)], dtype=np.uint64)

    def get(self, minimizer: int) -> int:
        return int(self._L.k2o_table_get(C.byref(self._db), minimizer))

    def scan(self, seq: bytes):
        n = max(0, len(seq) - self.k + 1)
        mins = np.zeros(max(n, 1), dtype=np.uint64)
        amb = np.zeros(max(n, 1), dtype=np.uint8)
        buf = np.frombuffer(seq, dtype=np.uint8) if len(seq) else np.zeros(1, np.uint8)
        got = self._L.k2o_scan_minimizers(C.byref(self._db), buf.ctypes.data, len(seq),
                                          mins.ctypes.data, amb.ctypes.data, n)
        assert got == n, (got, n)
        return mins[:n], amb[:n]

    def classify(self, bases: np.ndarray, seq_offsets: np.ndarray, paired: bool,
                 confidence: float = 0.0, *, want_taxa: bool = False, threads: int = 1):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        seq_offsets = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        n_seq = seq_offsets.size - 1
        mates = 2 if paired else 1
        assert n_seq % mates == 0
        n_frag = n_seq // mates
        out = np.zeros(n_frag, dtype=RESULT_DTYPE)
        lookups = np.zeros(n_frag, dtype=np.uint32)
        bptr = bases.ctypes.data if bases.size else None
        if threads > 1 and not want_taxa:
            rc = self._L.k2o_classify_mt(C.byref(self._db), bptr, seq_offsets.ctypes.data, n_frag,
                                         int(paired), float(confidence), out.ctypes.data,
                                         lookups.ctypes.data, threads)
            if rc != 0:
                raise RuntimeError(self._L.k2o_last_error().decode())
            return out, lookups
        taxa = taxa_off = None
        cap = 0
        if want_taxa:
            lens = np.diff(seq_offsets).astype(np.int64)
            cap = int(np.maximum(lens - self.k + 1, 0).sum() + (n_frag if paired else 0))
            taxa = np.zeros(max(cap, 1), dtype=np.uint32)
            taxa_off = np.zeros(n_frag + 1, dtype=np.uint64)
        rc = self._L.k2o_classify(C.byref(self._db), bptr, seq_offsets.ctypes.data, n_frag,
                                  int(paired), float(confidence), out.ctypes.data,
                                  lookups.ctypes.data,
                                  taxa.ctypes.data if want_taxa else None,
                                  taxa_off.ctypes.data if want_taxa else None, cap)
        if rc != 0:
            raise RuntimeError(self._L.k2o_last_error().decode())
        if want_taxa:
            return out, lookups, taxa[:cap], taxa_off
        return out, lookups


def pack_reads(reads, paired: bool = False):
    """reads: list of bytes (SE) or list of (bytes, bytes) (PE) -> (bases u8, seq_offsets u64)."""
    seqs = []
    for r in reads:
        if paired:
            seqs.extend(r)
        else:
            seqs.append(r)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return bases, offs
