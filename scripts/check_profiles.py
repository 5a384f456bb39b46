"""Bench hygiene: every PMC / kernel-stats summary under a profile directory must have been collected on the kernel the bench line
next to it names (`roofline.kernel` is what the library reports having launched, `cl_tuning.kernel_name`).

    python scripts/check_profiles.py <bench_line.json> <pmc_summary.json | kernel_stats.csv> [...]

Exit code 1 (and a message per mismatch) when a summary belongs to another kernel: stale counters must not sit beside a newer kernel."""
import csv
import json
import sys


def kernels_of(line_path):
    d = json.load(open(line_path))
    r = d['roofline']
    names = [k for k in (r.get('kernel') or '').split('+') if k]
    return names


def main():
    line, files = sys.argv[1], sys.argv[2:]
    want = kernels_of(line)
    bad = 0
    for f in files:
        if f.endswith('.json'):
            seen = json.load(open(f)).get('_kernel', {}).get('kernel', '')
            ok = any(k in seen for k in want)
        else:
            with open(f, newline='') as fh:
                names = [row.get('Name') or row.get('Kernel_Name') or '' for row in csv.DictReader(fh)]
            ok = all(any(k in n for n in names) for k in want)
            seen = f'{len(names)} kernels'
        print(('ok  ' if ok else 'MISMATCH') + f' {f}: bench line names {want}; summary has {seen[:160]}')
        bad += not ok
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
