R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  export VILF_SO=$R/tools/ab/$n.so
  rm -rf $R/gpurun_out/ps_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ps_$n -- python3 $R/bench.py --stress --stress-windows 1 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ps_$n.json 2> $R/gpurun_out/ps_$n.err
  python3 $R/tools/dev_summ.py $(find $R/gpurun_out/ps_$n -name '*kernel_stats.csv' | head -1) 8
done
