# quick GPU round: chosen tests, a bench line and the rocprofv3 kernel table of the same command. Usage: bash tools/gpu_quick.sh TAG "pytest args" ["bench args"]
T=${1:-q}; PT=${2:-tests/test_scan2map.py}; BA=${3:---steps 8 --warmup 2 --no-cpu-baseline --ragged-windows 0 --no-pcie}
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest $PT -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; tail -4 gpurun_out/$T/pytest.log
timeout -k 10 300 python bench.py $BA > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err || tail -c 1500 gpurun_out/$T/bench.err
python - <<P
import json
try:
    d=json.load(open("gpurun_out/$T/bench.json"))
    print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],3))
    k=d["roofline"]["kernels_ms_per_step"]; print({a:round(b,2) for a,b in k.items()})
    print("lidar", round(sum(v for a,v in k.items() if a.startswith("s2m")),2), "solve", round(k["k_linearize"]+k["k_solve"]+k["k_step"],2), "marg", round(sum(v for a,v in k.items() if "marg" in a or "prior" in a),2))
except Exception as e: print("bench parse failed", e)
P
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --ragged-windows 0 --no-pcie > $R/gpurun_out/$T/prof_bench.json 2> $R/gpurun_out/$T/prof.err
cd $R
F=$(find gpurun_out/$T/prof -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && cp $F gpurun_out/$T/kernel_stats.csv && python - <<P
import csv
rows=list(csv.DictReader(open("gpurun_out/$T/kernel_stats.csv")))
for r in rows[:28]: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f} tot_ms {float(r["TotalDurationNs"])/1e6:9.2f}')
P
