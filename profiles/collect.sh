#!/bin/bash
# Collects the rocprofv3 evidence for one bench.py configuration on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag> [bench.py args...]
# Writes under gpurun_out/prof_<tag>/: kernel trace + stats, and one --pmc pass per counter group (separate passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: SQ has 8 slots, FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).
# The program follows `--` directly (no env / bash -c hop).  Summaries are produced by profiles/summarize.py and copied
# into profiles/ by hand (gpurun_out/ is scratch).
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# per-kernel evidence is taken one batch at a time (--streams 1): with two streams a dispatch shares the chip and its
# duration / counters are not the kernel's own
args="--steps 20 --warmup 5 --cpu-samples 0 --no-breakdown --streams 1 $*"
echo "[collect] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py $args > $out/bench_trace.json 2> $out/trace.err
echo "[collect] pmc sq1"; rocprofv3 --kernel-trace --output-format csv -d $out/sq1 -o t --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM -- python3 bench.py $args > $out/bench_sq1.json 2> $out/sq1.err
echo "[collect] pmc sq2"; rocprofv3 --kernel-trace --output-format csv -d $out/sq2 -o t --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -- python3 bench.py $args > $out/bench_sq2.json 2> $out/sq2.err
echo "[collect] pmc fetch"; rocprofv3 --kernel-trace --output-format csv -d $out/fetch -o t --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python3 bench.py $args > $out/bench_fetch.json 2> $out/fetch.err
echo "[collect] pmc write"; rocprofv3 --kernel-trace --output-format csv -d $out/write -o t --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 bench.py $args > $out/bench_write.json 2> $out/write.err
echo "[collect] kernel trace, default scheduling (3 streams since round 3)"; rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace2 -o t -- python3 bench.py --steps 20 --warmup 5 --cpu-samples 0 --no-breakdown > $out/bench_trace2.json 2> $out/trace2.err
find $out -name '*.csv' | sort
