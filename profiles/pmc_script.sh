#!/bin/bash
# SQ counter passes for an arbitrary python script (kernel microbenchmarks): profiles/pmc_script.sh <tag> <script.py> [args]
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/sq1 -o t --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM -- python3 "$@" > $out/sq1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/sq2 -o t --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- python3 "$@" > $out/sq2.log 2>&1
find $out -name '*counter_collection.csv'
