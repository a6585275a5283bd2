#!/bin/bash
# quick SQ counter pass for k_classify on the bench workload:  gpurun -- 'bash scripts/pmc_quick.sh tag'
TAG=${1:-q}; REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY -d "$OUT/p1" -o p1 -- $BENCH > /dev/null 2> "$OUT/p1.err"
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_I8 -d "$OUT/p2" -o p2 -- $BENCH > /dev/null 2> "$OUT/p2.err"
cd "$REPO"; python3 scripts/summarize_prof.py "$OUT" | grep -A12 "k_classify" | cut -c1-110
