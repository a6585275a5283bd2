"""Engine: Python view of the C ABI (include/nohuman_engine.h).  numpy in, numpy out."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

RESULT_DTYPE = np.dtype([("call", "<u4"), ("total_kmers", "<u4"), ("clade_hits", "<u4"),
                         ("hit_groups", "<u4")])
TAXON_AMBIGUOUS = 0xFFFFFFFF
TAXON_MATE_BORDER = 0xFFFFFFFE
FLAG_PAIRED = 1
FLAG_LONG = 2


class EngineError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__("nohuman engine error %d: %s" % (code, message))
        self.code = code
        self.message = message


def _check(rc: int):
    if rc != 0:
        raise EngineError(rc, _lib.lib().nh_last_error().decode(errors="replace"))


def probe() -> str:
    """`nohuman --check` counterpart (/root/reference/src/lib.rs:50-57): raises if unusable."""
    buf = C.create_string_buffer(256)
    rc = _lib.lib().nh_probe(buf, 256)
    if rc != 0:
        raise EngineError(rc, buf.value.decode(errors="replace"))
    return buf.value.decode()


def run(db_dir, in1, out1, in2=None, out2=None, kraken_output=None, report=None, confidence: float = 0.0,
        threads: int = 1, keep_human: bool = False, device_ids=None, out_codec: int = 0,
        codec_threads: int = 0) -> "_lib.nh_stats":
    """nh_run: whole run, database loaded into every listed device (default: all visible)."""
    a = _lib.nh_run_args()
    a.db_dir = os.fsencode(db_dir)
    a.in1 = os.fsencode(in1)
    a.in2 = os.fsencode(in2) if in2 else None
    a.out1 = os.fsencode(out1)
    a.out2 = os.fsencode(out2) if out2 else None
    a.kraken_output = os.fsencode(kraken_output) if kraken_output else None
    a.report = os.fsencode(report) if report else None
    a.confidence = float(confidence)
    a.threads = int(threads)
    a.keep_human = int(bool(keep_human))
    a.out_codec = int(out_codec)
    a.codec_threads = int(codec_threads)
    ids = None
    if device_ids:
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        a.n_devices = len(device_ids)
        a.device_ids = ids
    else:
        a.n_devices = 0
        a.device_ids = None
    s = _lib.nh_stats()
    _check(_lib.lib().nh_run(C.byref(a), C.byref(s)))
    return s


def device_count() -> int:
    n = C.c_int(0)
    _check(_lib.lib().nh_device_count(C.byref(n)))
    return n.value


class Engine:
    """A kraken2 database resident in one GPU's HBM plus the classify entry points."""

    def __init__(self, handle):
        self._L = _lib.lib()
        self._h = handle

    # -- constructors -------------------------------------------------------------------------
    @classmethod
    def open(cls, db_dir, device: int = 0) -> "Engine":
        L = _lib.lib()
        h = C.c_void_p()
        _check(L.nh_open(os.fsencode(db_dir), device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_images(cls, opts: bytes, taxo: bytes, hashb, device: int = 0) -> "Engine":
        L = _lib.lib()
        h = C.c_void_p()
        hb = np.frombuffer(hashb, dtype=np.uint8) if isinstance(hashb, (bytes, bytearray)) \
            else np.ascontiguousarray(hashb).view(np.uint8)
        _check(L.nh_open_images(opts, len(opts), taxo, len(taxo), hb.ctypes.data, hb.nbytes, device,
                                C.byref(h)))
        return cls(h)

    @classmethod
    def synthetic(cls, capacity: int, n_keys: int, depth: int = 30, seed: int = 1,
                  device: int = 0) -> "Engine":
        L = _lib.lib()
        h = C.c_void_p()
        _check(L.nh_open_synthetic(capacity, n_keys, depth, seed, device, C.byref(h)))
        return cls(h)

    def close(self):
        if self._h is not None:
            self._L.nh_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # -- database ------------------------------------------------------------------------------
    @property
    def info(self) -> _lib.nh_db_info:
        i = _lib.nh_db_info()
        _check(self._L.nh_db_info_get(self._h, C.byref(i)))
        return i

    def db_check(self) -> _lib.nh_db_check:
        """What the content check of nh_open* found: non-empty cells, largest value, load factor, seconds."""
        c = _lib.nh_db_check()
        _check(self._L.nh_db_check_get(self._h, C.byref(c)))
        return c

    def reload_launch_knobs(self):
        """Tuning / test hook: NOHUMAN_FRAG_CHUNK / NOHUMAN_SEG_CAP / NOHUMAN_SCHED are read when an engine is opened;
        this reads them again (no launch may be in flight)."""
        fn = self._L.nh_debug_reload_knobs
        fn.restype = None
        fn.argtypes = [C.c_void_p]
        fn(self._h)

    def options(self) -> _lib.nh_options:
        o = _lib.nh_options()
        _check(self._L.nh_options_get(self._h, C.byref(o)))
        return o

    def set_options(self, *, minimum_hit_groups=None, linear_probing=None, reset_per_mate=None,
                    ambiguity_rule=None):
        o = self.options()
        if ambiguity_rule is not None:
            # the rule as a plain index (0 = last l-mer, 1 = mmscanner.h is_ambiguous()); the C ABI keeps 0 for
            # "the engine's default" (include/nohuman_engine.h: NH_AMBIGUITY_LAST_LMER 1, NH_AMBIGUITY_QUEUE 2)
            if int(ambiguity_rule) not in (0, 1):
                raise EngineError(-1, "ambiguity_rule must be 0 (last l-mer) or 1 (queue)")
            o.ambiguity_rule = int(ambiguity_rule) + 1
        if minimum_hit_groups is not None:
            o.minimum_hit_groups = int(minimum_hit_groups)
        if linear_probing is not None:
            o.linear_probing = int(linear_probing)
        if reset_per_mate is not None:
            o.reset_per_mate = int(reset_per_mate)
        _check(self._L.nh_options_set(self._h, C.byref(o)))

    def ambiguity_rule(self) -> int:
        """The rule in force as a plain index (0 = last l-mer, 1 = queue)"""
        return int(self.options().ambiguity_rule) - 1

    def external_id(self, internal: int) -> int:
        v = C.c_uint64(0)
        _check(self._L.nh_taxon_external(self._h, internal, C.byref(v)))
        return v.value

    def download_table(self) -> np.ndarray:
        cells = np.empty(self.info.capacity, dtype=np.uint32)
        _check(self._L.nh_table_download(self._h, cells.ctypes.data, cells.size))
        return cells

    def taxonomy_image(self) -> bytes:
        n = C.c_size_t(0)
        _check(self._L.nh_taxonomy_image(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(self._L.nh_taxonomy_image(self._h, buf, n.value, C.byref(n)))
        return buf.raw

    def opts_image(self) -> bytes:
        n = C.c_size_t(0)
        _check(self._L.nh_opts_image(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(self._L.nh_opts_image(self._h, buf, n.value, C.byref(n)))
        return buf.raw

    # -- classify ------------------------------------------------------------------------------
    def classify(self, bases: np.ndarray, seq_offsets: np.ndarray, paired: bool = False,
                 confidence: float = 0.0, want_taxa: bool = False, long_reads: bool = False):
        """Host buffers in, result records (RESULT_DTYPE) out; optional per-k-mer taxa list.  long_reads:
        NH_FLAG_LONG (the library sets it itself for batches of more than 2000 bases per fragment)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        seq_offsets = np.ascontiguousarray(seq_offsets, dtype=np.uint64)
        mates = 2 if paired else 1
        n_seq = seq_offsets.size - 1
        if n_seq % mates:
            raise ValueError("paired input needs an even number of sequences")
        n_frag = n_seq // mates
        flags = (FLAG_PAIRED if paired else 0) | (FLAG_LONG if long_reads else 0)
        out = np.zeros(n_frag, dtype=RESULT_DTYPE)
        bptr = bases.ctypes.data if bases.size else None
        if not want_taxa:
            _check(self._L.nh_classify_batch(self._h, bptr, seq_offsets.ctypes.data, n_frag, flags,
                                             float(confidence), out.ctypes.data, None, None, 0))
            return out
        cap = int(self._L.nh_kmer_taxa_entries(self._h, seq_offsets.ctypes.data, n_frag, flags))
        taxa = np.zeros(cap + 1, dtype=np.uint32)
        toff = np.zeros(n_frag + 1, dtype=np.uint64)
        _check(self._L.nh_classify_batch(self._h, bptr, seq_offsets.ctypes.data, n_frag, flags,
                                         float(confidence), out.ctypes.data, taxa.ctypes.data,
                                         toff.ctypes.data, cap + 1))
        return out, taxa[:cap], toff

    def classify_device(self, d_bases: int, d_seq_offsets: int, n_frag: int, paired: bool,
                        confidence: float, d_results: int, d_counters: int = 0, stream: int = 0,
                        d_kmer_taxa: int = 0, d_kmer_taxa_offsets: int = 0, long_reads: bool = False):
        """Device pointers in (ints), asynchronous on `stream` (a hipStream_t as int)."""
        _check(self._L.nh_classify_batch_device(
            self._h, d_bases, d_seq_offsets, n_frag,
            (FLAG_PAIRED if paired else 0) | (FLAG_LONG if long_reads else 0),
            float(confidence), d_results, d_kmer_taxa or None, d_kmer_taxa_offsets or None,
            d_counters or None, stream or None))

    def classify_records_device(self, d_text: int, text_len: int, d_seq_starts: int, d_seq_lens: int, n_frag: int,
                                paired: bool, confidence: float, d_results: int, d_counters: int = 0,
                                stream: int = 0, d_kmer_taxa: int = 0, d_kmer_taxa_offsets: int = 0,
                                long_reads: bool = False):
        """Sequences in place inside a device buffer of record text: (start, length) per sequence."""
        _check(self._L.nh_classify_records_device(
            self._h, d_text, text_len, d_seq_starts, d_seq_lens, n_frag,
            (FLAG_PAIRED if paired else 0) | (FLAG_LONG if long_reads else 0), float(confidence), d_results,
            d_kmer_taxa or None, d_kmer_taxa_offsets or None, d_counters or None, stream or None))

    def add_sequences(self, d_bases: int, d_seq_offsets: int, n_seq: int, value: int, stream: int = 0):
        """Bench/test support: insert the minimizers of device-resident sequences into the table."""
        _check(self._L.nh_synthetic_add_sequences(self._h, d_bases, d_seq_offsets, n_seq, value,
                                                  stream or None))

    def stats(self) -> _lib.nh_stats:
        s = _lib.nh_stats()
        _check(self._L.nh_stats_get(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        _check(self._L.nh_stats_reset(self._h))

    # -- whole run -----------------------------------------------------------------------------
    def run(self, in1, out1, in2=None, out2=None, kraken_output=None, report=None,
            confidence: float = 0.0, threads: int = 1, keep_human: bool = False, out_codec: int = 0,
            codec_threads: int = 0) -> _lib.nh_stats:
        a = _lib.nh_run_args()
        a.db_dir = None
        a.in1 = os.fsencode(in1)
        a.in2 = os.fsencode(in2) if in2 else None
        a.out1 = os.fsencode(out1)
        a.out2 = os.fsencode(out2) if out2 else None
        a.kraken_output = os.fsencode(kraken_output) if kraken_output else None
        a.report = os.fsencode(report) if report else None
        a.confidence = float(confidence)
        a.threads = int(threads)
        a.keep_human = int(bool(keep_human))
        a.n_devices = 1
        a.device_ids = None
        a.out_codec = int(out_codec)
        a.codec_threads = int(codec_threads)
        s = _lib.nh_stats()
        _check(self._L.nh_run_engine(self._h, C.byref(a), C.byref(s)))
        return s
