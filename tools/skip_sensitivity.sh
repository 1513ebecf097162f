# GPU box: end-to-end cost of every detect / match kernel with everything overlapped: an experiment build of the library
# (-DVO_EXPERIMENT_SKIP) leaves one kernel's launches out after the warm-up (stale but valid data flows on).
set -e
cd visual_slam_amd/csrc
mkdir -p ../../build/exp
F="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -DVO_EXPERIMENT_SKIP"
for f in vo_api vo_detect vo_pose vo_ba; do /opt/rocm/bin/hipcc $F -c $f.hip -o ../../build/exp/sk_$f.o & done
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form -c vo_match.hip -o ../../build/exp/sk_vo_match.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/exp/libvo_skip.so ../../build/exp/sk_*.o
cd ../..
for m in 0 1 2 4 8 16 32 64 128 256 512 1024 2048 0; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-extras --lib build/exp/libvo_skip.so --skip-mask $m > gpurun_out/skip_$m.json 2> gpurun_out/skip.err
  python -c "import json; d=json.load(open('gpurun_out/skip_$m.json')); print('skip mask',$m,'ms/step',d['ms_per_step'])"
done
