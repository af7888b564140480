# what do the window kernels' waves wait for? latency-level counters (SQ_INST_LEVEL_* / SQ_INSTS_* = mean latency in cycles), wait counters, L2 hit rate
# (rocprofv3 --pmc passes; counters only). Usage: bash tools/dev_pmc_wait.sh TAG
set -e
R=$PWD
T=${1:-wait}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/$T/counters.txt 2>&1 || true
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-stress-leg --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie --distinct-lidar 4 --no-lidar-stage"
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/$n -- python3 $R/bench.py $ARGS > $R/gpurun_out/$T/$n.json 2> $R/gpurun_out/$T/$n.err || { tail -5 $R/gpurun_out/$T/$n.err; }; }
run w1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
run w2 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES
run w3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run w4 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
run w5 SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
find $R/gpurun_out/$T -name '*counter_collection.csv' | head
