set -e
R=$PWD
mkdir -p $R/gpurun_out/lw
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lw/prof -- python3 $R/tools/dev_lw.py > $R/gpurun_out/lw/log.txt 2>&1
grep -v "^E2026\|^W2026" $R/gpurun_out/lw/log.txt | tail -3
f=$(find $R/gpurun_out/lw/prof -name '*kernel_stats.csv' | head -1)
head -14 $f | cut -c1-160
