for B in 1 4 8 16 32 48; do
  for mode in split one; do
    if [ $mode = one ]; then export VILF_NO_LIN_SPLIT=1; else unset VILF_NO_LIN_SPLIT; fi
    python bench.py --windows $B --distinct $B --no-lidar-stage --no-marginalize --no-cpu-baseline --no-pcie --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-stress-leg --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$B $mode', round(j['ms_per_step'],3), 'ms per solve of the batch;', round(j['value']), 'it/s; k_linearize per launch', round(j['roofline']['avg_launch_ms']*1e3,1), 'us')"
  done
done
