# GPU box: per-rank shares of a strong-scaling run x submissions in flight
set -e
for cfg in "512 1" "512 2" "512 3" "1024 2" "1024 3" "4096 2" "4096 3"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-cpu --no-extras --pairs $1 --in-flight $2 > gpurun_out/sw_p$1_f$2.json 2> gpurun_out/sw.err
  python -c "import json,sys; d=json.load(open('gpurun_out/sw_p$1_f$2.json')); print('pairs',$1,'in flight',$2,d['value'],d['ms_per_step'],d['config']['pairs_ok'])"
done
