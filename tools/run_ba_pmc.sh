cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
m=${1:-pose}
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/bapmc_sq_$m -o p -- python3 $R/tools/ba_bench.py 5 $m > $R/gpurun_out/bapmc_sq_$m.log 2>&1 || exit 1
timeout -s KILL 90 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/bapmc_tcc_$m -o p -- python3 $R/tools/ba_bench.py 5 $m > $R/gpurun_out/bapmc_tcc_$m.log 2>&1 || exit 1
