mkdir -p gpurun_out/bd
timeout -k 10 900 python bench.py > gpurun_out/bd/bench.json 2> gpurun_out/bd/bench.err; echo "bench rc $?"
python - <<P
import json
d = json.loads(open("gpurun_out/bd/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
print({k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_step"].items()})
for k in ("estimate_td_batch", "converging_batch", "stages_overlapped"):
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d.get(k, {}).items() if a != "what" and not isinstance(b, dict)})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
P
