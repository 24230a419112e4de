#!/bin/bash
# TCC (L2) counter passes for a python script: profiles/pmc_tcc.sh <tag> <script.py> [args]   (FETCH_SIZE takes 3 of 4 TCC slots)
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/fetch -o t --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python3 "$@" > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/write -o t --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 "$@" > $out/write.log 2>&1
find $out -name '*counter_collection.csv'
