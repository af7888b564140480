# same-box A/B of scan-to-map kernel variants: tools/ab/<name>.so for every name given (VILF_SO selects the library); prints the LiDAR groups of bench.py's line
for n in "$@"; do
  VILF_SO=$PWD/tools/ab/$n.so timeout -k 10 200 python bench.py --no-stress-leg --no-cpu-baseline --steps 5 --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { tail -3 gpurun_out/ab_$n.err; exit 1; }
  python - "$n" <<PY
import json, sys
n = sys.argv[1]
d = json.loads(open("gpurun_out/ab_%s.json" % n).read().strip().splitlines()[-1]); k = d["roofline"]["kernels_ms_per_step"]
print(n, "ms/step %.3f" % d["ms_per_step"], {g: round(v, 3) for g, v in k.items() if g.startswith("s2m")})
PY
done
