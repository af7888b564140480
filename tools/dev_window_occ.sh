# how much do the two window kernels lose with ONE workgroup per CU instead of two (unused extra dynamic LDS)? — the slope that says what a third would buy
for V in "0 0" "40000 0" "0 40000"; do
  set -- $V
  VILF_LIN_LDS_EXTRA=$1 VILF_SB_LDS_EXTRA=$2 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pcie --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-lidar-stage --no-marginalize > /tmp/wo.json 2>/dev/null
  python -c "
import json; d=json.loads(open('/tmp/wo.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('extra LDS lin $1 sb $2:', 'k_linearize', round(k['k_linearize'],3), 'k_solve', round(k['k_solve'],3), 'ms per launch')"
done
