# concurrent stress windows vs the number of hardware queues the HIP runtime multiplexes its streams onto (GPU_MAX_HW_QUEUES, default 4)
mkdir -p gpurun_out/stress_q
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q python3 bench.py --stress --stress-windows 8,16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/stress_q/q$q.json 2> gpurun_out/stress_q/q$q.err
  python - <<P
import json
d = json.loads(open("gpurun_out/stress_q/q$q.json").read().strip().splitlines()[-1])
print("queues $q", "value", d["value"], "ms", d["ms_per_step"], {k: v for k, v in d.items() if "concurrent" in k or "single" in k})
P
done
