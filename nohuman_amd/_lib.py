"""Loader of libnohuman_engine.so (ctypes).  Fails loudly if the library is missing."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NOHUMAN_ENGINE_LIB", os.path.join(_HERE, "libnohuman_engine.so"))
_LIB = None


class nh_result(C.Structure):
    _fields_ = [("call", C.c_uint32), ("total_kmers", C.c_uint32), ("clade_hits", C.c_uint32),
                ("hit_groups", C.c_uint32)]


class nh_stats(C.Structure):
    _fields_ = [("total_sequences", C.c_uint64), ("classified", C.c_uint64),
                ("unclassified", C.c_uint64), ("total_bases", C.c_uint64),
                ("table_lookups", C.c_uint64), ("seconds", C.c_double)]


class nh_db_info(C.Structure):
    _fields_ = [("k", C.c_uint64), ("l", C.c_uint64), ("spaced_seed_mask", C.c_uint64),
                ("toggle_mask", C.c_uint64), ("minimum_acceptable_hash_value", C.c_uint64),
                ("revcom_version", C.c_int32), ("dna_db", C.c_int32), ("capacity", C.c_uint64),
                ("size", C.c_uint64), ("key_bits", C.c_uint64), ("value_bits", C.c_uint64),
                ("node_count", C.c_uint64), ("device", C.c_int32), ("reserved", C.c_int32)]


class nh_db_check(C.Structure):
    _fields_ = [("non_empty_cells", C.c_uint64), ("max_value", C.c_uint64), ("load_factor", C.c_double),
                ("seconds", C.c_double)]


class nh_options(C.Structure):
    _fields_ = [("minimum_hit_groups", C.c_uint32), ("linear_probing", C.c_int32),
                ("reset_per_mate", C.c_int32), ("ambiguity_rule", C.c_int32)]


class nh_run_args(C.Structure):
    _fields_ = [("db_dir", C.c_char_p), ("in1", C.c_char_p), ("in2", C.c_char_p),
                ("out1", C.c_char_p), ("out2", C.c_char_p), ("kraken_output", C.c_char_p),
                ("report", C.c_char_p), ("confidence", C.c_double), ("threads", C.c_uint32),
                ("keep_human", C.c_int32), ("n_devices", C.c_int32),
                ("device_ids", C.POINTER(C.c_int32)), ("out_codec", C.c_int32), ("codec_threads", C.c_uint32)]


# every symbol include/nohuman_engine.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "nh_last_error": (C.c_char_p, []),
    "nh_abi_version": (C.c_int, []),
    "nh_probe": (C.c_int, [C.c_char_p, C.c_size_t]),
    "nh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "nh_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_P)]),
    "nh_open_images": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, C.c_int,
                                 C.POINTER(_P)]),
    "nh_open_synthetic": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int,
                                    C.POINTER(_P)]),
    "nh_synthetic_add_sequences": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, _P]),
    "nh_close": (C.c_int, [_P]),
    "nh_db_info_get": (C.c_int, [_P, C.POINTER(nh_db_info)]),
    "nh_db_check_get": (C.c_int, [_P, C.POINTER(nh_db_check)]),
    "nh_options_get": (C.c_int, [_P, C.POINTER(nh_options)]),
    "nh_options_set": (C.c_int, [_P, C.POINTER(nh_options)]),
    "nh_taxon_external": (C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint64)]),
    "nh_table_download": (C.c_int, [_P, _P, C.c_uint64]),
    "nh_taxonomy_image": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "nh_opts_image": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "nh_classify_batch": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_double, _P, _P, _P,
                                    C.c_uint64]),
    "nh_kmer_taxa_entries": (C.c_uint64, [_P, _P, C.c_uint64, C.c_uint32]),
    "nh_classify_batch_device": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_double, _P, _P,
                                           _P, _P, _P]),
    "nh_classify_records_device": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint64, C.c_uint32, C.c_double, _P,
                                             _P, _P, _P, _P]),
    "nh_stats_get": (C.c_int, [_P, C.POINTER(nh_stats)]),
    "nh_stats_reset": (C.c_int, [_P]),
    "nh_compress_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_uint32]),
    "nh_compress_file_device": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_int32]),
    "nh_gzip_gpu_file": (C.c_int, [C.c_int32, _P, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64)]),
    "nh_gunzip_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]),
    "nh_cache_bytes": (C.c_uint64, [C.c_int32, C.c_int32]),
    "nh_gunzip_device_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "nh_fastx_scan": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_uint64)]),
    "nh_run": (C.c_int, [C.POINTER(nh_run_args), C.POINTER(nh_stats)]),
    "nh_run_engine": (C.c_int, [_P, C.POINTER(nh_run_args), C.POINTER(nh_stats)]),
    "nh_allreduce_counters": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_uint64), C.c_char_p,
                                        C.c_size_t]),
}


def _preload_hip_runtime():
    """If PyTorch-ROCm is installed it bundles its own libamdhip64 (same SONAME as /opt/rocm's).
    Load that copy first so that this library and torch share ONE HIP runtime in the process;
    otherwise device pointers and streams could not be exchanged between the two."""
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    cand = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libnohuman_engine.so is not built (%s missing): run `make -C nohuman_amd/csrc` "
                "or __graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
