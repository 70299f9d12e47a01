"""python tools/kernel_avg.py KERNEL_STATS_CSV SUBSTRING: calls and average duration of the first kernel whose name contains SUBSTRING
(a rocprofv3 --stats kernel_stats.csv; kernel names may hold commas, hence a csv reader and not awk)."""
import csv
import sys
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Name"]:
        print(f'{row["Name"][:70]} calls {row["Calls"]} avg {float(row["AverageNs"]) / 1000:.1f} us')
        break
