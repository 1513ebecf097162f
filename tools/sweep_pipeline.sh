# GPU box: per-rank batch sizes of a strong-scaling run (4096 / N pairs per step), synchronous calls vs submit / wait
set -e
for cfg in "512 --sync-calls" "512" "1024 --sync-calls" "1024" "2048" "4096 --sync-calls" "4096"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-extras --pairs $1 $2 > gpurun_out/sw_p$1$2.json 2> gpurun_out/sw.err
  python -c "import json,sys; d=json.load(open('gpurun_out/sw_p$1$2.json')); print('pairs',$1,'$2',d['value'],d['ms_per_step'],d['config']['pairs_ok'])"
done
