# same-box A/B of library builds: tools/ab/<name>.so for every name given (VILF_SO selects the library; `cur` = the build in the tree); prints bench.py's per-step kernel groups
# usage: bash tools/dev_ab.sh <prefix of the groups to print, e.g. k_ or s2m> <name>...
P=$1; shift
for n in "$@"; do
  if [ "$n" = cur ]; then unset VILF_SO; else export VILF_SO=$PWD/tools/ab/$n.so; fi
  timeout -k 10 200 python bench.py --no-stress-leg --no-cpu-baseline --steps 5 --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { tail -3 gpurun_out/ab_$n.err; exit 1; }
  python - "$n" "$P" <<PY
import json, sys
n, p = sys.argv[1], sys.argv[2]
d = json.loads(open("gpurun_out/ab_%s.json" % n).read().strip().splitlines()[-1]); k = d["roofline"]["kernels_ms_per_step"]
print(n, "ms/step %.3f" % d["ms_per_step"], {g: round(v, 3) for g, v in k.items() if g.startswith(p)})
PY
done
