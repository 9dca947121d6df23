"""Print a kernel_stats.csv of rocprofv3 as a short table: python scratch/stats_table.py <csv> [rows]"""
import csv, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= n:
        break
    name = r['Name'].replace('tgru::', '')[:64]
    print(f"{name:66s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.2f} us  {float(r['Percentage']):5.1f}%")
