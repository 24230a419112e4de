set -e
mkdir -p gpurun_out/lines
O=gpurun_out/lines
python bench.py --steps 400 --warmup 50 > $O/r04_bf16_b256_bench_unprofiled.json 2> $O/err1.log
python bench.py --steps 20 --warmup 5 > $O/r04_bf16_b256_bench_driver_form.json 2>> $O/err1.log
python bench.py --steps 400 --warmup 50 --dtype fp8 > $O/r04_fp8_mid_b256_bench.json 2>> $O/err1.log
python bench.py --steps 400 --warmup 50 --dtype fp8 --fp8-policy wide > $O/r04_fp8_wide_b256_bench.json 2>> $O/err1.log
python bench.py --steps 400 --warmup 50 --dtype fp8 --fp8-policy all > $O/r04_fp8_all_b256_bench.json 2>> $O/err1.log
VV_NO_POS_TAIL=1 python bench.py --steps 400 --warmup 50 --cpu-samples 0 > $O/r04_ab_unfused_tail_bf16_b256_bench.json 2>> $O/err1.log
python bench.py --steps 400 --warmup 50 --cpu-samples 0 > $O/r04_ab_fused_tail_bf16_b256_bench.json 2>> $O/err1.log
echo 32done
python bench.py --steps 100 --warmup 20 --voxel 64 --batch 64 > $O/r04_bf16_d64_b64_bench.json 2>> $O/err1.log
python bench.py --steps 100 --warmup 20 --voxel 64 --batch 64 --dtype fp8 > $O/r04_fp8_mid_d64_b64_bench.json 2>> $O/err1.log
python bench.py --steps 100 --warmup 20 --voxel 64 --batch 64 --dtype fp8 --fp8-policy wide > $O/r04_fp8_wide_d64_b64_bench.json 2>> $O/err1.log
python bench.py --steps 100 --warmup 20 --voxel 64 --batch 64 --dtype fp8 --fp8-policy most > $O/r04_fp8_most_d64_b64_bench.json 2>> $O/err1.log
python bench.py --steps 100 --warmup 20 --voxel 64 --batch 64 --dtype fp8 --fp8-policy all > $O/r04_fp8_all_d64_b64_bench.json 2>> $O/err1.log
echo 64done
python bench.py --mode train --steps 100 --warmup 20 > $O/r04_train_bf16_b256_bench_unprofiled.json 2>> $O/err1.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 100 --warmup 20 > $O/r04_launcher_n1_eval.json 2>> $O/err1.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --mode train --steps 50 --warmup 10 > $O/r04_launcher_n1_train.json 2>> $O/err1.log
echo traindone
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -o t -- python3 bench.py --mode train --steps 20 --warmup 5 > $O/r04_train_bf16_b256_bench_under_rocprof.json 2> $O/train_trace.err
for f in $O/*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], '%.0f'%d['value'], '%.4f'%d['ms_per_step'])
PY
done
