#!/bin/bash
# Pins parity against the REAL kraken2 wherever one exists (it does not in the build container nor
# on the GPU box).  Runs the exact argv nohuman builds (/root/reference/src/main.rs:215-267) with
# the stock binary and the engine on the same database and inputs, then diffs
#   - the per-read kraken output (C/U, id, taxid, lengths, hit list)
#   - the kept-read FASTQ bytes
#   usage: scripts/parity_vs_kraken2.sh <db_dir> <reads_1.fq> [reads_2.fq] [confidence]
set -euo pipefail
DB=$1; IN1=$2; IN2=${3:-}; CONF=${4:-0}
command -v kraken2 >/dev/null || { echo "kraken2 not on PATH: parity vs kraken2 stays UNPINNED"; exit 3; }
REPO=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
if [ -n "$IN2" ]; then
  kraken2 --threads "$(nproc)" --db "$DB" --output "$T/ref.k" --confidence "$CONF" --paired \
          --unclassified-out "$T/ref_out#.fq" "$IN1" "$IN2" 2> "$T/ref.err"
  OUT="$T/eng_out#.fq"; ARGS=(--paired)
else
  kraken2 --threads "$(nproc)" --db "$DB" --output "$T/ref.k" --confidence "$CONF" \
          --unclassified-out "$T/ref_out.fq" "$IN1" 2> "$T/ref.err"
  OUT="$T/eng_out.fq"; ARGS=()
fi
PYTHONPATH="$REPO" python3 - "$DB" "$T/eng.k" "$CONF" "$OUT" "$IN1" ${IN2:+"$IN2"} <<'PY'
import sys
from nohuman_amd import CommandRunner
db, kout, conf, out, *inputs = sys.argv[1:]
argv = ["--threads", "1", "--db", db, "--output", kout, "--confidence", conf]
if len(inputs) == 2:
    argv.append("--paired")
argv += ["--unclassified-out", out] + inputs
CommandRunner("kraken2").run(argv)
PY
if diff -q "$T/ref.k" "$T/eng.k"; then
  echo "kraken output: IDENTICAL (the engine's default switches are pinned)"
else
  # self-diagnosing: the engine over the switch lattice (probing x per-mate reset x ambiguity rule x hit groups);
  # prints the ONE combination that reproduces kraken2's lines (tests/pin_lattice.py)
  (cd "$REPO" && python3 -m tests.pin_lattice "$DB" "$T/ref.k" "$CONF" "$IN1" ${IN2:+"$IN2"}) || true
fi
for f in "$T"/ref_out*.fq; do diff -q "$f" "${f/ref_out/eng_out}" && echo "$(basename "$f"): IDENTICAL"; done
grep -E "processed|classified" "$T/ref.err" || true
