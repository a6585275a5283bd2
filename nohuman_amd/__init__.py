"""nohuman_amd -- MI355X (gfx950) in-process replacement for nohuman's kraken2 subprocess.

Only the classification hot path lives here: HIP kernels + the C ABI of
include/nohuman_engine.h (csrc/), and a thin Python host mirror of the reference's boundary
(`CommandRunner`, /root/reference/src/lib.rs:11-58).  There is no CPU fallback: every entry fails
loudly when libnohuman_engine.so or a gfx950 device is missing.
"""
from .engine import Engine, EngineError, RESULT_DTYPE, probe, device_count  # noqa: F401
from .runner import CommandRunner, parse_kraken_stderr, validate_db_directory, \
    parse_confidence_score  # noqa: F401

__all__ = ["Engine", "EngineError", "RESULT_DTYPE", "probe", "device_count", "CommandRunner",
           "parse_kraken_stderr", "validate_db_directory", "parse_confidence_score"]
