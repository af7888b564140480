# the solve-only bench line (4096 windows, no LiDAR stage, no marginalization): k_linearize / k_solve_sb ms per launch — the quick A/B of a window-kernel change
python bench.py --no-lidar-stage --no-marginalize --no-cpu-baseline --no-pcie --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-stress-leg --steps 20 --warmup 3 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: round(v,4) for k,v in d['roofline']['kernels_ms'].items() if k in ('k_linearize','k_solve','k_step')})"
