# one GPU round: GPU tests, bench line, rocprofv3 kernel stats + the two PMC traffic passes of the same command, stress line. Usage: bash tools/gpu_round.sh TAG
set -e
T=${1:-r02}
mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/$T/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/$T/pytest_gpu.log
python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
python bench.py --stress --steps 5 --warmup 2 > gpurun_out/$T/bench_stress.json 2> gpurun_out/$T/bench_stress.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie --no-stress-leg > $R/gpurun_out/$T/prof_bench.json 2> $R/gpurun_out/$T/prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_fetch -- python3 $R/bench.py --distinct-lidar 4 --steps 2 --warmup 1 --no-cpu-baseline --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie --no-stress-leg > $R/gpurun_out/$T/pmc_fetch.json 2> $R/gpurun_out/$T/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_write -- python3 $R/bench.py --distinct-lidar 4 --steps 2 --warmup 1 --no-cpu-baseline --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie --no-stress-leg > $R/gpurun_out/$T/pmc_write.json 2> $R/gpurun_out/$T/pmc_write.err
find $R/gpurun_out/$T -name '*kernel_stats.csv' -o -name '*counter_collection.csv' | head
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-lidar-stage --no-marginalize --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie --no-stress-leg > $R/gpurun_out/$T/pmc_mfma.json 2> $R/gpurun_out/$T/pmc_mfma.err
find $R/gpurun_out/$T/pmc_mfma -name '*counter_collection.csv' | head -2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof_stress -- python3 $R/bench.py --stress --stress-windows 32 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/prof_stress.json 2> $R/gpurun_out/$T/prof_stress.err
cd $R
bash tools/dev_batch_sweep.sh $T > gpurun_out/$T/batch_sweep.log 2>&1 || true
