set -e
R=$PWD
mkdir -p $R/gpurun_out/r01c
python tools/dev_pcie.py > $R/gpurun_out/r01c/pcie.log 2>&1
cat $R/gpurun_out/r01c/pcie.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01c/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01c/pmc_fetch.json 2> $R/gpurun_out/r01c/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01c/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01c/pmc_write.json 2> $R/gpurun_out/r01c/pmc_write.err
find $R/gpurun_out/r01c/pmc_fetch $R/gpurun_out/r01c/pmc_write -name '*.csv' | head -20
