R=$PWD
mkdir -p $R/gpurun_out/lat
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lat/prof -- python3 $R/tools/dev_latency.py > $R/gpurun_out/lat/log.txt 2>&1
cd $R
python tools/dev_summ.py $(find gpurun_out/lat/prof -name '*kernel_stats.csv' | head -1) 12
