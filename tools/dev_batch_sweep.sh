# roofline fraction of the window solve vs the number of resident windows B (solve only, no LiDAR stage / marginalization)
set -e
T=${1:-sweep}
mkdir -p gpurun_out/$T
: > gpurun_out/$T/batch_sweep.jsonl
for B in 1 8 64 512 2048 4096; do
  D=$B; if [ $D -gt 64 ]; then D=64; fi
  python bench.py --windows $B --distinct $D --no-lidar-stage --no-marginalize --no-cpu-baseline --no-pcie --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-stress-leg --steps 20 --warmup 3 >> gpurun_out/$T/batch_sweep.jsonl 2>> gpurun_out/$T/batch_sweep.err
done
python - <<PY
import json
for l in open("gpurun_out/$T/batch_sweep.jsonl"):
    j = json.loads(l)
    print(j["config"].get("windows_per_gpu"), j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["achieved"])
PY
