set -e
R=$PWD
T=${1:-r01h}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lidar-stage --no-marginalize > $R/gpurun_out/$T/pmc_mfma.json 2> $R/gpurun_out/$T/pmc_mfma.err
find $R/gpurun_out/$T/pmc_mfma -name '*counter_collection.csv' | head -2
