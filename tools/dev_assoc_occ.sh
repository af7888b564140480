# is the association bound by memory latency? cap its workgroups per CU with unused dynamic LDS (VILF_ASSOC_LDS) and watch the group's time
for L in 0 40000 64000; do
  VILF_ASSOC_LDS=$L python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pcie --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-marginalize > /tmp/ao.json 2>/dev/null
  python -c "
import json; d=json.loads(open('/tmp/ao.json').read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms_per_step']; print('LDS cap $L', 's2m_associate', round(k['s2m_associate'],3), 'step', round(d['ms_per_step'],2))"
done
