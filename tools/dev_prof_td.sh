# per-kernel durations of the estimate_td batch (general path, 64 windows as one group)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ptd
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ptd -- python3 $R/tools/dev_prof_td.py 64 > $R/gpurun_out/ptd.log 2> $R/gpurun_out/ptd.err
tail -1 $R/gpurun_out/ptd.log
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/ptd/**/*kernel_stats.csv", recursive=True)[0]
r = list(csv.reader(open(f)))
for x in r[1:26]: print(x[0][:58].ljust(58), x[1].rjust(6), "avg %9.1f us" % (float(x[3]) / 1e3), "min %8.1f" % (float(x[5]) / 1e3), x[4].rjust(7), "%")
PY
