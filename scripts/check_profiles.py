"""Bench hygiene: every PMC / kernel-stats summary under a profile directory must have been collected on the kernel the bench line
next to it names (`roofline.kernel` is what the library reports having launched, `cl_tuning.kernel_name`).

    python scripts/check_profiles.py [--duration-tol 0.05] [--entry metric_shape] <bench_line.json> <pmc_summary.json | kernel_stats.csv> [...]

`--entry NAME` checks the sub-object `roofline.NAME` instead of `roofline` itself (round 6: the headline line's `roofline` is the HBM-true
17 x 1 048 576 launch, `roofline.metric_shape` the cache-resident 17 x 65 536 launch the value is timed on).

Exit code 1 (and a message per mismatch) when a summary belongs to another kernel: stale counters must not sit beside a newer kernel.
A line whose `roofline.traffic_source` names a summary file must also carry THAT file's traffic ((2 x FETCH_SIZE + WRITE_SIZE) KiB): a
summary re-collected after the line was written (round 4: 73.29 MB in the line, 68.84 MB in the summary it cited) fails here.
With --duration-tol F a kernel_stats.csv must also AGREE IN TIME: the rocprofv3 `AverageNs` of the kernel a single-kernel line names
has to lie within F (relative) of that line's `roofline.launch_us` (HIP events inside bench.py) -- the two clocks behind `roofline.frac`."""
import csv
import json
import sys


ENTRY = None


def roofline_of(line_path):
    r = json.load(open(line_path))['roofline']
    return r[ENTRY] if ENTRY else r


def kernels_of(line_path):
    r = roofline_of(line_path)
    names = [k for k in (r.get('kernel') or '').split('+') if k]
    return names


def main():
    argv = sys.argv[1:]
    tol = None
    global ENTRY
    while argv and argv[0] in ('--duration-tol', '--entry'):
        if argv[0] == '--duration-tol':
            tol, argv = float(argv[1]), argv[2:]
        else:
            ENTRY, argv = argv[1], argv[2:]
    line, files = argv[0], argv[1:]
    want = kernels_of(line)
    roof = roofline_of(line)
    launch_us = roof.get('launch_us')
    bad = 0
    for f in files:
        if f.endswith('.json'):
            summary = json.load(open(f))
            seen = summary.get('_kernel', {}).get('kernel', '')
            ok = any(k in seen for k in want)
            import os
            if ok and roof.get('traffic') is not None and roof.get('traffic_source') == os.path.basename(f) and 'FETCH_SIZE' in summary and 'WRITE_SIZE' in summary:
                in_file = (2.0 * summary['FETCH_SIZE']['mean'] + summary['WRITE_SIZE']['mean']) * 1024.0
                ok = abs(in_file - roof['traffic']) <= 1e-3 * in_file
                seen += f"; line traffic {roof['traffic'] / 1e6:.2f} MB vs {in_file / 1e6:.2f} MB in the summary it names"
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
