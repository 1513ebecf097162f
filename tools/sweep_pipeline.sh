# GPU box: contexts x lanes x pose-batch x hardware-queue sweep of bench.py (A/B of the in-context concurrency)
set -e
for cfg in "1 4 100000 4" "1 4 100000 8" "1 3 100000 4" "1 5 100000 8" "1 6 100000 8" "1 8 100000 8" "1 4 1024 4" "1 4 1024 8" "1 4 2048 8" "1 4 512 8" "2 4 100000 8" "1 2 100000 4"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$4 VO_LANES=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-extras --streams $1 --pose-batch $3 > gpurun_out/sw_$1_$2_$3_$4.json 2> gpurun_out/sw.err
  python -c "import json,sys; d=json.load(open('gpurun_out/sw_$1_$2_$3_$4.json')); print('contexts',$1,'lanes',$2,'pose_batch',$3,'hwq',$4,d['value'],d['config']['pairs_ok'])"
done
