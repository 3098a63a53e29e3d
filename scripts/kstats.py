#!/usr/bin/env python3
"""print rocprofv3 kernel stats per step: python scripts/kstats.py <dir> <iters>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
it = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%-100s %6.1f/step %8.1f us  %8.2f us/step" % (r['Name'][:100], int(r['Calls']) / it, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / it / 1e3))
print("total us/step %.1f, launches/step %.1f" % (sum(float(r['TotalDurationNs']) for r in rows) / it / 1e3, sum(int(r['Calls']) for r in rows) / it))
