set -e
R=$PWD
mkdir -p $R/gpurun_out/s2b
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s2b/prof -- python3 $R/tools/dev_s2b.py 2048 > $R/gpurun_out/s2b/log.txt 2>&1
tail -5 $R/gpurun_out/s2b/log.txt
f=$(find $R/gpurun_out/s2b/prof -name '*kernel_stats.csv' | head -1)
head -30 $f
