"""Host-side mirror of the reference's boundary around the hot path.

`CommandRunner` keeps the name, arguments and error behaviour of the reference type
(/root/reference/src/lib.rs:11-58) but `run()` hands the kraken2 argv that nohuman builds
(/root/reference/src/main.rs:210-267) to the in-process engine through the C ABI (`nh_run`)
instead of spawning `kraken2`.  The small validators beside it mirror src/lib.rs:61-151 so the
parity tests read like the reference's own (src/lib.rs:153-222).
"""
from __future__ import annotations

import logging
import os
from pathlib import Path

from . import _lib
from .engine import Engine, EngineError, probe

log = logging.getLogger("nohuman")

REQUIRED_DB_FILES = ("hash.k2d", "opts.k2d", "taxo.k2d")


def parse_kraken_stderr(stderr: str):
    """(total, classified, unclassified) from kraken2's stderr summary
    (/root/reference/src/lib.rs:61-97; grammar SURVEY.md A.8).  Raises ValueError where the
    reference returns ParseIntError."""
    total = classified = unclassified = 0

    def first_int(line: str) -> int:
        parts = line.split()
        tok = (parts[0] if parts else "0").replace(",", "")
        if not tok.isascii() or not (tok.isdigit() or (tok[:1] == "+" and tok[1:].isdigit())):
            raise ValueError("invalid digit found in string")
        return int(tok)

    for line in stderr.splitlines():
        if "processed" in line:
            total = first_int(line)
        elif "sequences classified" in line:
            classified = first_int(line)
        elif "sequences unclassified" in line:
            unclassified = first_int(line)
    return total, classified, unclassified


def check_path_exists(s) -> Path:
    """/root/reference/src/lib.rs:100-107"""
    p = Path(s)
    if p.exists():
        return p
    raise ValueError('"%s" does not exist' % s)


def validate_db_directory(path) -> Path:
    """/root/reference/src/lib.rs:119-141: the directory, or its `db` subdirectory, must hold the
    three .k2d files."""
    path = Path(path)
    if path.is_dir() and all((path / f).exists() for f in REQUIRED_DB_FILES):
        return path
    sub = path / "db"
    if sub.is_dir() and all((sub / f).exists() for f in REQUIRED_DB_FILES):
        return sub
    raise ValueError("Required files (%s) not found in \"%s\" or its 'db' subdirectory"
                     % (", ".join(REQUIRED_DB_FILES), path))


def parse_confidence_score(s: str) -> float:
    """/root/reference/src/lib.rs:145-151: a number in the closed interval [0, 1]."""
    try:
        c = float(s)
    except ValueError:
        raise ValueError("Confidence score must be a number")
    if not (0.0 <= c <= 1.0):  # also rejects NaN, as RangeInclusive::contains does
        raise ValueError("Confidence score must be in the closed interval [0, 1]")
    return c


def _pct(a: int, b: int) -> str:
    # the reference prints f64 a/b*100 with {:.2}; 0/0 is NaN there (src/lib.rs:42)
    return "NaN" if b == 0 else "%.2f" % (a / b * 100.0)


class CommandRunner:
    """Drop-in for the reference's CommandRunner (src/lib.rs:11-58) over the GPU engine."""

    def __init__(self, command: str = "kraken2", device: int = 0):
        self.command = command
        self.device = device
        self.last_stats = None

    def is_executable(self) -> bool:
        """src/lib.rs:50-57: true iff the dependency is usable (here: library + gfx950 device)."""
        try:
            probe()
            return True
        except (EngineError, RuntimeError, OSError):
            return False

    @staticmethod
    def parse_argv(args):
        """The kraken2 options nohuman emits (src/main.rs:215-267) -> dict."""
        opts = {"threads": 1, "db": None, "output": None, "confidence": 0.0, "report": None,
                "paired": False, "classified_out": None, "unclassified_out": None, "inputs": []}
        it = iter(args)
        for a in it:
            if a == "--threads":
                opts["threads"] = int(next(it))
            elif a == "--db":
                opts["db"] = next(it)
            elif a == "--output":
                opts["output"] = next(it)
            elif a == "--confidence":
                opts["confidence"] = parse_confidence_score(next(it))
            elif a == "--report":
                opts["report"] = next(it)
            elif a == "--paired":
                opts["paired"] = True
            elif a == "--classified-out":
                opts["classified_out"] = next(it)
            elif a == "--unclassified-out":
                opts["unclassified_out"] = next(it)
            elif a.startswith("--"):
                raise OSError("%s failed with stderr Unknown option: %s" % ("kraken2", a))
            else:
                opts["inputs"].append(a)
        return opts

    def run_stock(self, args) -> None:
        """The reference's own path, verbatim (src/lib.rs:22-48): spawn `self.command` with the argv,
        fail on a non-zero exit with its stderr, scrape the three summary integers.  Only taken when
        NOHUMAN_STOCK_KRAKEN2=1 (BASELINE.json configs[0]; the parity pin against a real kraken2)."""
        import subprocess
        from types import SimpleNamespace
        try:
            out = subprocess.run([self.command] + [str(a) for a in args], capture_output=True)
        except OSError as e:  # Command::output()? -> io::Error (e.g. binary not found)
            raise OSError(str(e)) from e
        stderr_log = out.stderr.decode("utf-8", errors="replace")
        if out.returncode != 0:
            raise OSError("%s failed with stderr %s" % (self.command, stderr_log))
        log.debug("kraken2 stderr:\n %s", stderr_log)
        try:
            total, classified, unclassified = parse_kraken_stderr(stderr_log)
        except ValueError:  # .unwrap_or((0, 0, 0))
            total = classified = unclassified = 0
        self.last_stats = SimpleNamespace(total_sequences=total, classified=classified, unclassified=unclassified)
        log.info("%d / %d (%s%%) sequences classified as human; %d (%s%%) as non-human",
                 classified, total, _pct(classified, total), unclassified, _pct(unclassified, total))

    def run(self, args) -> None:
        """src/lib.rs:22-48.  Raises OSError("<command> failed with stderr ...") on failure and
        logs the reference's summary line on success."""
        if os.environ.get("NOHUMAN_STOCK_KRAKEN2") == "1":
            return self.run_stock(args)
        o = self.parse_argv(args)
        try:
            if o["db"] is None:
                raise EngineError(-1, "--db is required")
            n_in = len(o["inputs"])
            if n_in not in (1, 2) or (o["paired"] != (n_in == 2)):
                raise EngineError(-1, "--paired requires exactly two inputs")
            out = o["classified_out"] or o["unclassified_out"]
            if out is None:
                raise EngineError(-1, "--classified-out or --unclassified-out is required")
            if o["paired"]:
                if "#" not in out:
                    raise EngineError(-1, "paired filename format missing # character: %s" % out)
                out1, out2 = out.replace("#", "_1", 1), out.replace("#", "_2", 1)
            else:
                out1, out2 = out, None
            a = _lib.nh_run_args()
            a.db_dir = os.fsencode(o["db"])
            a.in1 = os.fsencode(o["inputs"][0])
            a.in2 = os.fsencode(o["inputs"][1]) if n_in == 2 else None
            a.out1 = os.fsencode(out1)
            a.out2 = os.fsencode(out2) if out2 else None
            a.kraken_output = os.fsencode(o["output"]) if o["output"] else None
            a.report = os.fsencode(o["report"]) if o["report"] else None
            a.confidence = o["confidence"]
            a.threads = o["threads"]
            a.keep_human = 1 if o["classified_out"] else 0
            a.n_devices = 1
            import ctypes as C
            dev = (C.c_int32 * 1)(self.device)
            a.device_ids = dev
            s = _lib.nh_stats()
            L = _lib.lib()
            rc = L.nh_run(C.byref(a), C.byref(s))
            if rc != 0:
                raise EngineError(rc, L.nh_last_error().decode(errors="replace"))
        except EngineError as e:
            raise OSError("%s failed with stderr %s" % (self.command, e.message)) from e
        self.last_stats = s
        log.info("%d / %d (%s%%) sequences classified as human; %d (%s%%) as non-human",
                 s.classified, s.total_sequences, _pct(s.classified, s.total_sequences),
                 s.unclassified, _pct(s.unclassified, s.total_sequences))
