set -e
mkdir -p gpurun_out/r01c
python -m pytest tests -m gpu -x -q > gpurun_out/r01c/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r01c/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r01c/pytest_gpu.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r01c/bench.json 2> gpurun_out/r01c/bench.err
cat gpurun_out/r01c/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c/prof -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01c/prof_bench.json 2> $R/gpurun_out/r01c/prof.err
cat $R/gpurun_out/r01c/prof_bench.json
find $R/gpurun_out/r01c/prof -name '*kernel_stats.csv' | head
