# GPU box: A/B of library builds (build/exp/libvo_*.so) through bench.py, two rounds
set -e
for rep in 1 2; do for v in s8 s3 s4 s2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-extras --lib build/exp/libvo_$v.so > gpurun_out/sw_$v.json 2> gpurun_out/sw.err
  python -c "import json,sys; d=json.load(open('gpurun_out/sw_$v.json')); print('$v',d['value'],d['config']['pairs_ok'], d['kernels']['k_select#1'])"
done; done
