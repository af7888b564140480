# instruction-cache behaviour of the window kernels (rocprofv3 --pmc, counters only): SQC_ICACHE_* and the fetch level. usage: bash tools/dev_pmc_icache.sh TAG
set -e
R=$PWD
T=${1:-icache}
mkdir -p $R/gpurun_out/$T
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-stress-leg --ragged-windows 0 --converging-windows 0 --td-windows 0 --no-latency --no-pcie --distinct-lidar 4 --no-lidar-stage"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/i1 -- python3 $R/bench.py $ARGS > $R/gpurun_out/$T/i1.json 2> $R/gpurun_out/$T/i1.err || tail -5 $R/gpurun_out/$T/i1.err
cd $R
python3 - <<P
import csv, collections, glob
f=glob.glob('gpurun_out/$T/i1/*/*counter_collection.csv')[0]
vals=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    vals[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k in ('k_linearize','k_solve_sb','k_marg_prepare','k_marg_schur','k_mf_chol'):
    d=vals.get(k)
    if not d: continue
    def top(n):
        v=sorted(d.get(n,[0])); v=v[len(v)*3//4:] or [0]; return sum(v)/len(v)
    print(k, {n: '%.4g'%top(n) for n in sorted(d)})
P
