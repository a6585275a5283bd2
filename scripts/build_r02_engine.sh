#!/bin/bash
# Builds the round-2 kernels (commit 0a1d40a: round-2 device code + round-3 host fixes) as tools/old_engine_r02.so, the
# reference side of tools/ab_r02_r03.sh (same-box comparisons; the .so is git-ignored and travels with gpurun).
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$REPO" archive 0a1d40a nohuman_amd/csrc include | tar -x -C "$TMP"
make -s -C "$TMP/nohuman_amd/csrc" ../libnohuman_engine.so
cp "$TMP/nohuman_amd/libnohuman_engine.so" "$REPO/tools/old_engine_r02.so"
rm -rf "$TMP"
ls -la "$REPO/tools/old_engine_r02.so"
