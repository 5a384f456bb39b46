"""Summarise rocprofv3 output directories into small JSON / CSV files for profiles/.

    python scripts/pmc_summary.py <out.json> <kernel-substring> <counter_collection.csv> [<counter_collection.csv> ...]

Per counter: dispatches, mean / min / max of the per-dispatch value for kernels whose name contains the substring;
only the dispatches with the LARGEST grid of that kernel are kept (benchmark shape, not the small parity launches).
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    out, needle, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(list)
    meta = {}
    rows = []
    for f in files:
        with open(f, newline='') as fh:
            rows += [row for row in csv.DictReader(fh) if needle in row['Kernel_Name']]
    biggest = max(int(row['Grid_Size']) for row in rows)
    for row in rows:
        if int(row['Grid_Size']) == biggest:
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
                meta = {'kernel': row['Kernel_Name'], 'grid': row['Grid_Size'], 'workgroup': row['Workgroup_Size'],
                        'vgpr': row['VGPR_Count'], 'sgpr': row['SGPR_Count'], 'lds': row['LDS_Block_Size']}
    summary = {k: {'dispatches': len(v), 'mean': sum(v) / len(v), 'min': min(v), 'max': max(v)} for k, v in acc.items()}
    summary['_kernel'] = meta
    with open(out, 'w') as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))


if __name__ == '__main__':
    main()
