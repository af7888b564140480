# per-kernel device time of an estimate_td batch of 64 windows (the general path as one group)
R=$PWD
mkdir -p gpurun_out/td_trace
cd /tmp && export TMPDIR=/tmp
VILF_TD_B=64 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/td_trace/prof -- python3 $R/tools/dev_td_batch.py > $R/gpurun_out/td_trace/out.txt 2> $R/gpurun_out/td_trace/err.txt
cd $R
python - <<P
import csv, glob, collections, re
f = glob.glob("gpurun_out/td_trace/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def nm(s):
    s = s.replace("(anonymous namespace)::","").replace("void ","")
    return re.split(r"[(<]", s)[0][:40]
rows = [r for r in rows if nm(r["Kernel_Name"]).startswith("lw_") or "k_imu_prep" in r["Kernel_Name"]]
# the last solve: the last 1/7 of the lw launches
n = len(rows) // 7
last = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))[-n:]
A = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = nm(r["Kernel_Name"]); A[k][0] += 1; A[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in A.values()); span = max(int(r["End_Timestamp"]) for r in last) - min(int(r["Start_Timestamp"]) for r in last)
print("launches", len(last), "sum of durations ms", tot / 1e6, "span ms", span / 1e6)
for k, (c, t) in sorted(A.items(), key=lambda x: -x[1][1]): print("%-22s %4d %8.1f us  avg %6.1f" % (k, c, t / 1e3, t / c / 1e3))
P
