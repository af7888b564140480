# same-box A/B of the LiDAR stage only (no window solve beside it: --steps 20 of the scan-to-map step, 8 distinct scenes), variants interleaved and repeated:
# bash tools/dev_ab_lidar.sh REPEATS name... (`cur` = the build in the tree). b_map_update's time moves by +-5 % between processes on one box; compare medians.
R=$1; shift
for r in $(seq 1 $R); do
for n in "$@"; do
  if [ "$n" = cur ]; then unset VILF_SO; else export VILF_SO=$PWD/tools/ab/$n.so; fi
  timeout -k 10 200 python bench.py --no-stress-leg --no-cpu-baseline --steps 20 --warmup 2 --distinct-lidar 8 --distinct 8 --no-marginalize --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie > gpurun_out/abl_$n.json 2> gpurun_out/abl_$n.err || { tail -3 gpurun_out/abl_$n.err; exit 1; }
  python - "$n" <<PY
import json, sys
n = sys.argv[1]
d = json.loads(open("gpurun_out/abl_%s.json" % n).read().strip().splitlines()[-1]); k = d["roofline"]["kernels_ms_per_step"]
print(n, "ms/step %.3f" % d["ms_per_step"], {g[4:]: round(v, 3) for g, v in k.items() if g.startswith("s2m") and v > 0.1})
PY
done
done
