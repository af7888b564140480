# is the concurrent stress solve (S handles side by side) bound by the launch rate or by the GPU? kernel trace of S = 8: busy time of the union of kernel intervals vs wall
R=$PWD
mkdir -p gpurun_out/stress_trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/stress_trace/prof -- python3 $R/bench.py --stress --stress-windows 8 --steps 4 --warmup 1 --no-cpu-baseline > $R/gpurun_out/stress_trace/out.json 2> $R/gpurun_out/stress_trace/err.txt
cd $R
python - <<P
import csv, glob
f = glob.glob("gpurun_out/stress_trace/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
# the last 60 % of the trace = the S = 8 phase (after warm-ups and the single-window loop)
t0, t1 = iv[0][0], max(e for _, e in iv)
lo = t0 + int(0.55 * (t1 - t0))
iv = [(s, e) for s, e in iv if s >= lo]
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
wall = max(e for _, e in iv) - iv[0][0]
tot = sum(e - s for s, e in iv)
print("kernels", len(iv), "wall ms", wall / 1e6, "union busy ms", busy / 1e6, "sum of durations ms", tot / 1e6, "launches/s", len(iv) / (wall / 1e9))
P
