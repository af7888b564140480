# which pipe is busy? SQ counters of every kernel of the default bench (three rocprofv3 --pmc passes; counters only, no other trace domain)
set -e
R=$PWD
T=${1:-pipes}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-stress-leg --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie --distinct-lidar 4"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/p1 -- python3 $R/bench.py $ARGS > $R/gpurun_out/$T/p1.json 2> $R/gpurun_out/$T/p1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/p2 -- python3 $R/bench.py $ARGS > $R/gpurun_out/$T/p2.json 2> $R/gpurun_out/$T/p2.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/p3 -- python3 $R/bench.py $ARGS > $R/gpurun_out/$T/p3.json 2> $R/gpurun_out/$T/p3.err
find $R/gpurun_out/$T -name '*counter_collection.csv' | head -5
