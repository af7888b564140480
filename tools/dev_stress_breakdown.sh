# per-kernel time of the configs[4] group solve at S windows (default 32): rocprofv3 kernel trace of `bench.py --stress --stress-windows S`, the kernels of the LAST 45 % of the
# trace (the S-window phase) summed by name.  usage: bash tools/dev_stress_breakdown.sh [S] [tag]
S=${1:-32}; TAG=${2:-s$S}
R=$PWD
mkdir -p gpurun_out/stress_bd_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/stress_bd_$TAG/prof -- python3 $R/bench.py --stress --stress-windows $S --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/stress_bd_$TAG/out.json 2> $R/gpurun_out/stress_bd_$TAG/err.txt
cd $R
python3 - <<P
import csv, glob, json, collections, re
f = glob.glob("gpurun_out/stress_bd_$TAG/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# phases of bench.py --stress: warm-up single solves, the S-window group (1 warm-up + 3 timed), then 3 profiled single solves. Split by the grid z dimension = windows in the group.
def z(r):
    for k in ("Grid_Size_Z", "Grid_Size_z"):
        if k in r: return int(r[k])
    return 1
grp = [r for r in rows if z(r) == $S] if $S > 1 else rows
acc = collections.defaultdict(lambda: [0, 0.0])
for r in grp:
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    acc[n][0] += 1; acc[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
solves = 4.0 if $S > 1 else None
tot = sum(v[1] for v in acc.values())
wall = (int(grp[-1]["End_Timestamp"]) - int(grp[0]["Start_Timestamp"])) / 1e6
out = {"windows": $S, "kernels_ms_per_group_solve": {k: v[1] / (solves or 1) for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])}, "launches_per_group_solve": {k: v[0] / (solves or 1) for k, v in acc.items()},
       "sum_ms_per_group_solve": tot / (solves or 1), "first_to_last_ms": wall}
json.dump(out, open("gpurun_out/stress_bd_$TAG/breakdown.json", "w"), indent=1)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-28s %6d launches %9.3f ms per group solve" % (k[:28], v[0], v[1] / (solves or 1)))
print("sum", tot / (solves or 1), "ms per group solve; bench:", open("gpurun_out/stress_bd_$TAG/out.json").read()[:300])
P
