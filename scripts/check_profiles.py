"""Bench hygiene: every PMC / kernel-stats summary under a profile directory must have been collected on the kernel the bench line
next to it names (`roofline.kernel` is what the library reports having launched, `cl_tuning.kernel_name`).

    python scripts/check_profiles.py [--duration-tol 0.05] <bench_line.json> <pmc_summary.json | kernel_stats.csv> [...]

Exit code 1 (and a message per mismatch) when a summary belongs to another kernel: stale counters must not sit beside a newer kernel.
With --duration-tol F a kernel_stats.csv must also AGREE IN TIME: the rocprofv3 `AverageNs` of the kernel a single-kernel line names
has to lie within F (relative) of that line's `roofline.launch_us` (HIP events inside bench.py) -- the two clocks behind `roofline.frac`."""
import csv
import json
import sys


def kernels_of(line_path):
    d = json.load(open(line_path))
    r = d['roofline']
    names = [k for k in (r.get('kernel') or '').split('+') if k]
    return names


def main():
    argv = sys.argv[1:]
    tol = None
    if argv and argv[0] == '--duration-tol':
        tol, argv = float(argv[1]), argv[2:]
    line, files = argv[0], argv[1:]
    want = kernels_of(line)
    launch_us = json.load(open(line))['roofline'].get('launch_us')
    bad = 0
    for f in files:
        if f.endswith('.json'):
            seen = json.load(open(f)).get('_kernel', {}).get('kernel', '')
            ok = any(k in seen for k in want)
        else:
            with open(f, newline='') as fh:
                rows = list(csv.DictReader(fh))
            names = [row.get('Name') or row.get('Kernel_Name') or '' for row in rows]
            ok = all(any(k in n for n in names) for k in want)
            seen = f'{len(names)} kernels'
            if ok and tol is not None and len(want) == 1 and launch_us:
                row = max((r for r in rows if want[0] in (r.get('Name') or r.get('Kernel_Name') or '')), key=lambda r: int(r['Calls']))
                avg_us = float(row['AverageNs']) / 1e3
                rel = abs(avg_us - launch_us) / launch_us
                ok = rel <= tol
                seen += f"; rocprofv3 average {avg_us:.2f} us over {row['Calls']} dispatches vs launch_us {launch_us:.2f} us in the line: {rel * 100:.1f} % apart (limit {tol * 100:.0f} %)"
        print(('ok  ' if ok else 'MISMATCH') + f' {f}: bench line names {want}; summary has {seen[:260]}')
        bad += not ok
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
