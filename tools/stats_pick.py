"""Print calls / average / total of the kernels whose name contains one of the given substrings, from a rocprofv3
kernel_stats.csv.  Usage: python tools/stats_pick.py <kernel_stats.csv> substr [substr ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r["Name"] for k in sys.argv[2:]):
        print("%-64s x%-5s avg %9.1f us  total %8.2f ms  %5s%%" % (r["Name"].replace("void (anonymous namespace)::", "")[:64],
              r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
