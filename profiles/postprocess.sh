#!/bin/bash
# Turns gpurun_out/prof_<tag>/ (written by profiles/collect.sh on the GPU box) into the summaries committed under profiles/.
#   profiles/postprocess.sh r02_bf16 r02
set -e
tag=$1; out=$2
P=gpurun_out/prof_$tag
python profiles/summarize.py bygrid $P/trace/t_kernel_trace.csv > profiles/${out}_bf16_b256_kernel_by_grid.csv
cp $P/trace/t_kernel_stats.csv profiles/${out}_bf16_b256_kernel_stats.csv
cp $P/trace2/t_kernel_stats.csv profiles/${out}_bf16_b256_default_streams_kernel_stats.csv
python profiles/summarize.py counters $P/sq1/t_counter_collection.csv $P/sq2/t_counter_collection.csv $P/fetch/t_counter_collection.csv $P/write/t_counter_collection.csv > profiles/${out}_bf16_b256_counters.csv
python profiles/summarize.py traffic $P/fetch/t_counter_collection.csv $P/write/t_counter_collection.csv \
  'E1=first_conv_chain_kernel:262144' 'E2=conv_direct16_kernel:262144' 'E3=sd_kernel<0>:131072' 'E4=pg_kernel<0>:114688' 'E4sum_E5=lt_e5x_kernel:65536' \
  'D2=pg_kernel<1>:114688' 'D3=sd_kernel<1>:131072' 'D4=ctw16_kernel:131072' 'D5=final_bce_sweepw_kernel:262144' > profiles/${out}_pmc_traffic.json
cp $P/bench_trace.json profiles/${out}_bf16_b256_bench_under_rocprof.json
cp $P/bench_trace2.json profiles/${out}_bf16_b256_default_streams_bench_under_rocprof.json
