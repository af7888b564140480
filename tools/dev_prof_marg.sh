set -e
R=$PWD
mkdir -p $R/gpurun_out/mg
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mg/prof -- python3 $R/tools/dev_marg_batch.py > $R/gpurun_out/mg/log.txt 2>&1
f=$(find $R/gpurun_out/mg/prof -name '*kernel_stats.csv' | head -1)
grep -E "k_mf|k_marg|k_prior" $f | cut -d, -f1-7
