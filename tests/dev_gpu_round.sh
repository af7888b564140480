set -e
mkdir -p gpurun_out/r01f
python -m pytest tests -m gpu -x -q > gpurun_out/r01f/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r01f/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r01f/pytest_gpu.log
python bench.py > gpurun_out/r01f/bench.json 2> gpurun_out/r01f/bench.err
cat gpurun_out/r01f/bench.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01f/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01f/prof_bench.json 2> $R/gpurun_out/r01f/prof.err
cat $R/gpurun_out/r01f/prof_bench.json
find $R/gpurun_out/r01f/prof -name '*kernel_stats.csv' | head
