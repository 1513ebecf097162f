# GPU box: the rocprofv3 evidence of a build.  usage: bash tools/profile_r2.sh TAG   -> gpurun_out/TAG_*
# Each pass is a run of its own (counters are never combined with tracing; one --pmc set per run), one kernel at a time.
set -e
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --pairs 256 --workers 1 --steps 2 --warmup 1 --no-cpu --no-extras --streams 1 --chunk 32"
VO_SERIAL=1 VO_LANES=1 timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -o kt --output-format csv -- $BENCH > gpurun_out/${TAG}_kt_bench.json 2> gpurun_out/${TAG}_kt.err
PMCB="python3 bench.py --pairs 64 --workers 1 --steps 1 --warmup 0 --no-cpu --no-extras --streams 1 --chunk 32"
VO_SERIAL=1 VO_LANES=1 timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/${TAG}_pmc_valu -o p --output-format csv -- $PMCB > /dev/null 2> gpurun_out/${TAG}_pmc_valu.err
VO_SERIAL=1 VO_LANES=1 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_pmc_fetch -o p --output-format csv -- $PMCB > /dev/null 2> gpurun_out/${TAG}_pmc_fetch.err
VO_SERIAL=1 VO_LANES=1 timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc_write -o p --output-format csv -- $PMCB > /dev/null 2> gpurun_out/${TAG}_pmc_write.err
find gpurun_out/${TAG}_kt gpurun_out/${TAG}_pmc_valu gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write -name "*.csv" | head -20
