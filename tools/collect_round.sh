# copy what tools/gpu_round.sh TAG left under gpurun_out/TAG into profiles/ under PREFIX (e.g. r05_v2). Usage: bash tools/collect_round.sh TAG PREFIX [WINDOWS]
set -e
T=$1; P=profiles/$2; B=${3:-4096}; G=gpurun_out/$T
ks=$(find $G/prof -name '*kernel_stats.csv' | head -1); cp "$ks" ${P}_kernel_stats.csv
ks=$(find $G/prof_stress -name '*kernel_stats.csv' | head -1); [ -n "$ks" ] && cp "$ks" ${P}_stress_kernel_stats.csv
grep '^{' $G/bench.json | tail -1 > ${P}_bench.json
grep '^{' $G/bench_stress.json | tail -1 > ${P}_bench_stress.json
cp $G/batch_sweep.jsonl ${P}_batch_sweep.jsonl
f=$(find $G/pmc_fetch -name '*counter_collection.csv' | head -1); w=$(find $G/pmc_write -name '*counter_collection.csv' | head -1)
python3 profiles/summarize_pmc.py "$f" "$w" 3 $P $B lidar+solve+marginalize
m=$(find $G/pmc_mfma -name '*counter_collection.csv' | head -1)
python3 profiles/summarize_pmc_mfma.py "$m" ${P}_pmc_mfma.json $B
tail -1 $G/pytest_gpu.log
