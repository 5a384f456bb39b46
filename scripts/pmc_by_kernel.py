"""Mean per-dispatch value of every counter in rocprofv3 counter_collection.csv files, grouped by (kernel, grid, workgroup):
    python scripts/pmc_by_kernel.py <needle> <csv> [<csv> ...]"""
import csv, json, sys
from collections import defaultdict
needle, files = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    with open(f, newline='') as fh:
        for row in csv.DictReader(fh):
            if needle in row['Kernel_Name']:
                key = (row['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', ''), row['Grid_Size'], row['Workgroup_Size'], row['VGPR_Count'])
                acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
for key, counters in sorted(acc.items()):
    print(json.dumps({'kernel': key[0], 'grid': int(key[1]), 'workgroup': int(key[2]), 'vgpr': int(key[3]),
                      **{c: round(sum(v) / len(v), 1) for c, v in counters.items()}, 'dispatches': len(next(iter(counters.values())))}))
