set -e
T=${1:-r01g}
mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/$T/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/$T/pytest_gpu.log
python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/prof_bench.json 2> $R/gpurun_out/$T/prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/pmc_fetch.json 2> $R/gpurun_out/$T/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$T/pmc_write.json 2> $R/gpurun_out/$T/pmc_write.err
find $R/gpurun_out/$T -name '*kernel_stats.csv' -o -name '*counter_collection.csv' | head
