import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 22]:
    n=r['Name']
    if 'onesweep_iteration' in n: n='rocprim onesweep_iteration'
    elif 'onesweep_global_offsets' in n: n='rocprim onesweep_histogram/offsets'
    elif 'scan_impl' in n: n='rocprim scan'
    elif 'rocprim' in n: n='rocprim other'
    print(f"{n[:58]:58s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} pct {100*float(r['TotalDurationNs'])/tot:5.1f}")
