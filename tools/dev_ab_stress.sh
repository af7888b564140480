for n in "$@"; do
  VILF_SO=$PWD/tools/ab/$n.so timeout -k 10 200 python bench.py --stress --stress-windows 1 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/abs_$n.json 2> gpurun_out/abs_$n.err || { tail -3 gpurun_out/abs_$n.err; exit 1; }
  python - "$n" <<PY
import json, sys
n = sys.argv[1]
d = json.loads(open("gpurun_out/abs_%s.json" % n).read().strip().splitlines()[-1])
c = d["concurrent_windows"][0]
print(n, "ms per solve %.3f" % c["ms_per_solve_of_all_windows"], c["kernels_ms_per_group_solve"])
PY
done
