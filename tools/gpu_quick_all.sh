# all GPU tests in one process, then the stress bench
mkdir -p gpurun_out/quick
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/quick/tests.txt 2>&1; echo "tests rc $?"; tail -3 gpurun_out/quick/tests.txt
timeout -k 10 300 python bench.py --stress --steps 5 --warmup 2 > gpurun_out/quick/stress.json 2> gpurun_out/quick/stress.err; echo "stress rc $?"
python - <<P
import json
d = json.loads(open("gpurun_out/quick/stress.json").read().strip().splitlines()[-1])
print("value", d["value"], "single", d.get("single_window"), d["config"]["parallelism"])
for r in d.get("concurrent_windows", []): print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
print(d["roofline"]["kernels_ms_per_solve"], d["cpu_baseline"])
P
