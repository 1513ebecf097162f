# GPU box: the rocprofv3 evidence of the round-3 build.  usage: bash tools/profile_r3.sh TAG   -> gpurun_out/TAG_*
# Each pass is a run of its own (counters are never combined with tracing; one --pmc set per run), one kernel at a time.
set -e
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export VO_SERIAL=1 VO_LANES=1
B="python3 $R/bench.py --workers 1 --no-cpu --no-extras --streams 1"
O=$R/gpurun_out
# headline frame size (1241x376, 3000 features), 64 frames per launch
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt -o kt --output-format csv -- $B --pairs 256 --steps 2 --warmup 1 --chunk 32 > $O/${TAG}_kt_bench.json 2> $O/${TAG}_kt.err
P="$B --pairs 64 --steps 1 --warmup 0 --chunk 32"
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/${TAG}_pmc_valu -o p --output-format csv -- $P > /dev/null 2> $O/${TAG}_pmc_valu.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o p --output-format csv -- $P > /dev/null 2> $O/${TAG}_pmc_fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o p --output-format csv -- $P > /dev/null 2> $O/${TAG}_pmc_write.err
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY -d $O/${TAG}_pmc_lds -o p --output-format csv -- $P > /dev/null 2> $O/${TAG}_pmc_lds.err
# configs[3]: 3840x2160, 20000 features, 16 frames per launch
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt4k -o kt --output-format csv -- $B --config4 --pairs 16 --steps 1 --warmup 1 --chunk 8 > $O/${TAG}_kt4k_bench.json 2> $O/${TAG}_kt4k.err
# configs[4]: the BA pass
unset VO_SERIAL VO_LANES
A="python3 $R/tools/ba_bench.py 20 all"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ba_kt -o kt --output-format csv -- $A > $O/${TAG}_ba_kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_ba_fetch -o p --output-format csv -- $A > $O/${TAG}_ba_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_ba_write -o p --output-format csv -- $A > $O/${TAG}_ba_write.log 2>&1
find $O/${TAG}_* -name "*.csv" | head -40
