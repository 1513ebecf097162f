# GPU box: lanes x chunk sweep of bench.py, two runs each (one context per GPU)
set -e
for rep in 1 2; do for cfg in "4 32" "6 32" "8 32" "4 16" "8 16" "4 64" "6 64" "6 24"; do
  set -- $cfg
  VO_LANES=$1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-extras --chunk $2 > gpurun_out/sw_l$1_c$2.json 2> gpurun_out/sw.err
  python -c "import json,sys; d=json.load(open('gpurun_out/sw_l$1_c$2.json')); print('lanes',$1,'chunk',$2,d['value'],d['config']['pairs_ok'])"
done; done
